"""Drop-in for ``util/util.py``: optimizer / scheduler factories (host-side torch.optim objects
over the generator's arena-backed parameters), ``tensor2im``, ``save_result``."""
from pathlib import Path

import os
import numpy as np
import torch
from torch.optim import lr_scheduler


# ---- scheduler / optimizer factories -------------------------------------------------------------------------------------
# One entry per policy the reference's config accepts (util/util.py:8-39).  Each builder gets the optimizer and the keyword
# arguments of get_scheduler and returns a torch.optim.lr_scheduler object with the reference's hyper-parameters.
def _linear_decay(opt, n_epochs_decay=None, **_):
    span = float(n_epochs_decay + 1)
    return lr_scheduler.LambdaLR(opt, lr_lambda=lambda epoch: max(1.0 - max(0, epoch) / span, 0))


_SCHEDULERS = {
    'linear': _linear_decay,
    'step': lambda opt, lr_decay_iters=None, **_: lr_scheduler.StepLR(opt, step_size=lr_decay_iters, gamma=0.5),
    'plateau': lambda opt, **_: lr_scheduler.ReduceLROnPlateau(opt, mode='min', factor=0.2, threshold=0.01, patience=5),
    'cosine': lambda opt, n_epochs=None, **_: lr_scheduler.CosineAnnealingLR(opt, T_max=n_epochs, eta_min=0),
    'none': lambda opt, **_: lr_scheduler.LambdaLR(opt, lr_lambda=lambda _epoch: 1),
}
_OPTIMIZERS = {
    'adam': lambda cfg, params: torch.optim.Adam(params, lr=cfg['lr'], betas=(cfg['optimizer_beta1'], cfg['optimizer_beta2'])),
    'rmsprop': lambda cfg, params: torch.optim.RMSprop(params, lr=cfg['lr']),
    'sgd': lambda cfg, params: torch.optim.SGD(params, lr=cfg['lr']),
}


def get_scheduler(optimizer, lr_policy, n_epochs=None, n_epochs_decay=None, lr_decay_iters=None):
    """``util/util.py:8-25``.  An unknown policy RETURNS (does not raise) a NotImplementedError, as the reference does."""
    build = _SCHEDULERS.get(lr_policy)
    if build is None:
        return NotImplementedError('learning rate policy [%s] is not implemented', lr_policy)
    return build(optimizer, n_epochs=n_epochs, n_epochs_decay=n_epochs_decay, lr_decay_iters=lr_decay_iters)


def get_optimizer(cfg, params):
    """``util/util.py:28-39``; same return-the-exception convention for unknown names."""
    build = _OPTIMIZERS.get(cfg['optimizer'])
    if build is None:
        return NotImplementedError('optimizer [%s] is not implemented', cfg['optimizer'])
    return build(cfg, params)


def _to_uint8_hwc(chw):
    """[3,H,W] float tensor in (roughly) [0,1] -> HxWx3 array scaled to 0..255 (still float)."""
    return chw.detach().clamp(0.0, 1.0).cpu().float().numpy().transpose(1, 2, 0) * 255.0


def tensor2im(input_image, imtype=np.uint8):
    """``util/util.py:42-52``: first image of a batch tensor -> HxWx3 array of ``imtype``; arrays are only cast, anything
    that is neither a tensor nor an array is handed back untouched."""
    if isinstance(input_image, np.ndarray):
        return input_image.astype(imtype)
    if not isinstance(input_image, torch.Tensor):
        return input_image
    return _to_uint8_hwc(input_image[0]).astype(imtype)


def save_result(image_t, dataroot):
    """``util/util.py:55-59``: writes ``<dataroot>/out/output.png`` (ToPILImage semantics: [3,H,W] float in [0,1] -> uint8)."""
    from PIL import Image
    arr = _to_uint8_hwc(image_t).astype(np.uint8)
    path = Path(f"{dataroot}/out")
    path.mkdir(exist_ok=True, parents=True)
    Image.fromarray(arr).save(f"{path}/output.png")


class AsyncResultWriter:
    """``save_result`` off the critical path.

    The reference's ``train.py:70-76`` stalls the loop every ``log_images_freq`` steps on a device->host copy plus
    a PNG encode.  ``submit`` only enqueues an asynchronous copy into a pinned host buffer on the current stream and
    an event; a worker thread waits for the event, converts (ToPILImage semantics) and writes
    ``<dataroot>/out/output.png``.  At most one image is in flight per slot (two slots): a newer result simply
    overwrites the file later, as the reference does.  ``close()`` drains the queue (called at the end of
    ``train_model`` so the final PNG is on disk when it returns)."""

    def __init__(self, dataroot, slots=2):
        import queue
        import threading
        self.dir = Path(f"{dataroot}/out")
        self.dir.mkdir(exist_ok=True, parents=True)
        self._q = queue.Queue()
        self._free = queue.Queue()
        for _ in range(slots):
            self._free.put(None)
        self._err = None
        # Round 5: out/output.png is OVERWRITTEN by every logged image (util/util.py:55-59 called from train.py:75), so only the latest one matters to a
        # reader.  At ~270 steps/s a logged image every 10 steps is 27 PNG encodes per second on a thread that shares the GIL with the launch loop
        # (measured: 8.1 -> 7.6 s per 2000-step pair without them).  Intermediate images closer than min_interval seconds are skipped; the caller forces the
        # last one.  SPLICE_LOG_MIN_INTERVAL=0 writes every logged image.
        self.min_interval = float(os.environ.get("SPLICE_LOG_MIN_INTERVAL", "0.25"))
        # (_last is written by the submitting thread only, _gap by the writer thread only and read by submit(): single float stores, atomic under the
        # interpreter lock; a stale _gap only moves ONE skip decision.  A run killed mid-training leaves a file at most max(min_interval, _gap) old.)
        self._last = -1e9
        self._gap = 0.0         # 10 x the duration of the last convert + encode + write (a 1200 x 900 image takes ~60 ms: the writer thread stays under 10 % duty)
        self.skipped = 0
        self._t = threading.Thread(target=self._run, name="splice-result-writer", daemon=True)
        self._t.start()

    def submit(self, image_t, force=True):
        """force=False: the image may be skipped when the previous one was submitted less than ``min_interval`` seconds ago (the file is
        overwritten every time anyway; the encoder thread shares the interpreter lock with the loop that launches the steps)."""
        import time
        now = time.monotonic()
        if not force and now - self._last < max(self.min_interval, self._gap):
            self.skipped += 1
            return
        self._last = now
        buf = self._free.get()                      # blocks only if `slots` images are still being written
        img = image_t.detach()
        if buf is None or buf.shape != img.shape:
            buf = torch.empty(img.shape, dtype=torch.float32, pin_memory=img.is_cuda)
        buf.copy_(img, non_blocking=True)
        ev = None
        if img.is_cuda:
            ev = torch.cuda.Event()
            ev.record()
        self._q.put((buf, ev))

    def _run(self):
        from PIL import Image
        while True:
            item = self._q.get()
            if item is None:
                return
            buf, ev = item
            try:
                if ev is not None:
                    ev.synchronize()
                import time
                t0 = time.monotonic()
                arr = (buf.clamp(0.0, 1.0).numpy().transpose(1, 2, 0) * 255.0).astype(np.uint8)
                tmp = self.dir / "output.png.tmp"
                Image.fromarray(arr).save(tmp, format="PNG")
                tmp.replace(self.dir / "output.png")      # readers never see a half-written file
                if self.min_interval > 0:
                    self._gap = 10.0 * (time.monotonic() - t0)
            except Exception as e:                         # surfaced by close()
                self._err = e
            self._free.put(buf)

    def close(self):
        self._q.put(None)
        self._t.join()
        if self._err is not None:
            raise self._err
