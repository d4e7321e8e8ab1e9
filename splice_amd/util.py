"""Drop-in for ``util/util.py``: optimizer / scheduler factories (host-side torch.optim objects
over the generator's arena-backed parameters), ``tensor2im``, ``save_result``."""
from pathlib import Path

import numpy as np
import torch
from torch.optim import lr_scheduler


def get_scheduler(optimizer, lr_policy, n_epochs=None, n_epochs_decay=None, lr_decay_iters=None):
    """``util/util.py:8-25``.  Unknown policies RETURN NotImplementedError, as the reference does."""
    if lr_policy == 'linear':
        def lambda_rule(epoch):
            lr_l = 1.0 - max(0, epoch) / float(n_epochs_decay + 1)
            return max(lr_l, 0)
        scheduler = lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda_rule)
    elif lr_policy == 'step':
        scheduler = lr_scheduler.StepLR(optimizer, step_size=lr_decay_iters, gamma=0.5)
    elif lr_policy == 'plateau':
        scheduler = lr_scheduler.ReduceLROnPlateau(optimizer, mode='min', factor=0.2, threshold=0.01, patience=5)
    elif lr_policy == 'cosine':
        scheduler = lr_scheduler.CosineAnnealingLR(optimizer, T_max=n_epochs, eta_min=0)
    elif lr_policy == 'none':
        scheduler = lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda x: 1)
    else:
        return NotImplementedError('learning rate policy [%s] is not implemented', lr_policy)
    return scheduler


def get_optimizer(cfg, params):
    """``util/util.py:28-39``."""
    if cfg['optimizer'] == 'adam':
        optimizer = torch.optim.Adam(params, lr=cfg['lr'], betas=(cfg['optimizer_beta1'], cfg['optimizer_beta2']))
    elif cfg['optimizer'] == 'rmsprop':
        optimizer = torch.optim.RMSprop(params, lr=cfg['lr'])
    elif cfg['optimizer'] == 'sgd':
        optimizer = torch.optim.SGD(params, lr=cfg['lr'])
    else:
        return NotImplementedError('optimizer [%s] is not implemented', cfg['optimizer'])
    return optimizer


def tensor2im(input_image, imtype=np.uint8):
    """``util/util.py:42-52``."""
    if not isinstance(input_image, np.ndarray):
        if isinstance(input_image, torch.Tensor):
            image_tensor = input_image.data
        else:
            return input_image
        image_numpy = image_tensor[0].clamp(0.0, 1.0).cpu().float().numpy()
        image_numpy = np.transpose(image_numpy, (1, 2, 0)) * 255.0
    else:
        image_numpy = input_image
    return image_numpy.astype(imtype)


def save_result(image_t, dataroot):
    """``util/util.py:55-59``: writes ``<dataroot>/out/output.png`` (ToPILImage semantics: [3,H,W] float in [0,1] -> uint8)."""
    from PIL import Image
    arr = (image_t.detach().clamp(0.0, 1.0).cpu().float().numpy().transpose(1, 2, 0) * 255.0).astype(np.uint8)
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
        self._t = threading.Thread(target=self._run, name="splice-result-writer", daemon=True)
        self._t.start()

    def submit(self, image_t):
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
                arr = (buf.clamp(0.0, 1.0).numpy().transpose(1, 2, 0) * 255.0).astype(np.uint8)
                tmp = self.dir / "output.png.tmp"
                Image.fromarray(arr).save(tmp, format="PNG")
                tmp.replace(self.dir / "output.png")      # readers never see a half-written file
            except Exception as e:                         # surfaced by close()
                self._err = e
            self._free.put(buf)

    def close(self):
        self._q.put(None)
        self._t.join()
        if self._err is not None:
            raise self._err
