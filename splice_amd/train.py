"""Drop-in for ``train.py``: ``train_model(dataroot, callback=None)``.

Same contract as the reference (``train.py:15-80``): reads ``conf/default/config.yaml`` from the
working directory (falling back to the packaged copy), seeds python / numpy / torch, opens the first
image of ``<dataroot>/A`` and ``<dataroot>/B``, runs ``n_epochs`` optimisation steps and every
``log_images_freq`` steps writes ``<dataroot>/out/output.png`` (asynchronously: ``util.AsyncResultWriter``) and calls
``callback(output[0])`` with the ``[3,H,W]`` float image.  The step itself is the fused HIP engine (``SpliceEngine``), one host
call per step, losses read back only when a progress line is printed.

Data feed: the reference augments PIL images on the CPU every step (``data/Dataset.py:62-70``).
Here the images live on the GPU and the same pipelines run on device tensors (``splice_amd/augment.py``):
structure image = h-flip(0.5), ColorJitter(.4,.4,.2,.1)@0.5, GaussianBlur(3)@0.2; texture image = h-flip(0.5);
then one random square crop covering >= ``min_cover`` of the height (``data/transforms.py:7-41``).
"""
import os
import random
from argparse import ArgumentParser

import numpy as np
import torch
import yaml

from . import augment
from .engine import SpliceEngine
from .util import AsyncResultWriter

device = torch.device('cuda' if torch.cuda.is_available() else 'cpu')
_PKG_CFG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "conf", "default", "config.yaml")



def fp8_mode(cfg):
    """Engine ``fp8`` argument of the config key ``fp8`` (one meaning everywhere: ``splice_amd.vit.fp8_mode``): False -- bf16;
    True / ``"gemm"`` -- e4m3 operands for the QKV / fc1 / fc2 projections and the self-similarity Gram matrices (the fastest
    measured setting); ``"attention"`` / ``"all"`` -- the attention forward in e4m3 as well (BASELINE configs[4] as written)."""
    from .vit import fp8_mode as _mode
    return _mode(cfg.get('fp8', False))


def _load_image(path, resize):
    from PIL import Image
    img = Image.open(path).convert('RGB')
    if resize and resize > 0:   # transforms.Resize(int) on a PIL image: shorter edge -> resize, bilinear
        w, h = img.size
        short, long = (w, h) if w <= h else (h, w)
        if short != resize:
            ns, nl = resize, int(resize * long / short)
            img = img.resize((ns, nl) if w <= h else (nl, ns), Image.BILINEAR)
    arr = np.asarray(img, dtype=np.float32) / 255.0          # ToTensor
    return torch.from_numpy(arr).permute(2, 0, 1).contiguous()


def _first_file(d):
    return os.path.join(d, os.listdir(d)[0])


_VIT_ENGINES = {}


def _release_process_caches():
    """Interpreter exit: free the process-wide ViT engine (weights, contexts, captured graphs) while the interpreter and the HIP
    runtime are still whole -- left to module teardown, its destructor runs in an arbitrary order against `_lib` and torch."""
    try:
        _VIT_ENGINES.clear()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
    except Exception:
        pass


import atexit
atexit.register(_release_process_caches)


def _shared_vit_engine(model_name, who):
    """The frozen DINO ViT of this process: loaded / packed once per (weight source, model) and shared by every run that follows
    (a batch worker optimises many pairs; reading and packing 86 M parameters per pair was 2 s of a 12 s run).  Weight
    source: ``SPLICE_DINO_CHECKPOINT=<dino .pth>`` or ``SPLICE_SYNTHETIC_WEIGHTS=1`` (seeded stand-in weights)."""
    from .vit import VitEngine
    ckpt = os.environ.get("SPLICE_DINO_CHECKPOINT")
    if ckpt:
        key = ("checkpoint", os.path.abspath(ckpt), os.path.getmtime(ckpt), model_name)
    elif os.environ.get("SPLICE_SYNTHETIC_WEIGHTS") == "1":
        key = ("synthetic", model_name)
    else:
        raise RuntimeError(f"{who}: no DINO weights (set SPLICE_DINO_CHECKPOINT=<dino .pth>, pass vit_state=..., "
                           "or SPLICE_SYNTHETIC_WEIGHTS=1); the reference's torch.hub download is not available here")
    if key not in _VIT_ENGINES:
        if ckpt:
            from .checkpoint import load_dino_checkpoint
            _, state = load_dino_checkpoint(ckpt, model_name)
        else:
            from . import synth
            state = synth.vit_params(1234, model_name, img_size=224)
        _VIT_ENGINES.clear()   # one resident weight set per process
        _VIT_ENGINES[key] = VitEngine(model_name, device=device).load_state_dict(state)
    return _VIT_ENGINES[key]


class DeviceDataFeed:
    """GPU-side counterpart of ``SingleImageDataset`` (data/Dataset.py:12-73)."""

    def __init__(self, cfg, A, B):
        self.cfg, self.A, self.B = cfg, A.to(device), B.to(device)
        self.step = -1

    def get_A(self):
        return self.A[None]

    @staticmethod
    def _crops(img, min_cover, n_crops):
        """``[n_crops,3,s,s]``: one size per call, one position per crop (data/transforms.py:19-27)."""
        _, h, w = img.shape
        size, boxes = augment.global_crop_boxes(h, w, min_cover, n_crops)
        return torch.stack([img[:, top:top + size, left:left + size] for top, left in boxes]).contiguous()

    def next(self):
        self.step += 1
        aug = bool(self.cfg['use_augmentations'])
        sample = {'step': self.step}
        if self.step % self.cfg['entire_A_every'] == 0:
            sample['A'] = self.get_A()
        # data/Dataset.py:67-68: augment the whole image, then crop
        A = augment.structure_transforms(self.A) if aug else self.A
        sample['A_global'] = self._crops(A, self.cfg['global_A_crops_min_cover'], self.cfg['global_A_crops_n_crops'])
        B = augment.texture_transforms(self.B) if aug else self.B
        sample['B_global'] = self._crops(B, self.cfg['global_B_crops_min_cover'], self.cfg['global_B_crops_n_crops'])
        return sample


def train_model(dataroot, callback=None, cfg_overrides=None, vit_state=None, progress=True):
    cfg_path = "conf/default/config.yaml" if os.path.exists("conf/default/config.yaml") else _PKG_CFG
    with open(cfg_path, "r") as f:
        cfg = yaml.safe_load(f)
    if dataroot is not None:
        cfg['dataroot'] = dataroot
    cfg.update(cfg_overrides or {})
    n_crops = (int(cfg['global_A_crops_n_crops']), int(cfg['global_B_crops_n_crops']))
    if not all(1 <= n <= 8 for n in n_crops):
        raise NotImplementedError("the fused engine takes 1..8 global crops per image (global_{A,B}_crops_n_crops)")
    if n_crops[0] == n_crops[1]:
        n_crops = n_crops[0]   # (unequal counts: the reference zips the crop lists, util/losses.py:76,87,98 -- so does the engine)
    if device.type != 'cuda':
        raise RuntimeError("train_model needs an MI355X: the product path has no CPU fallback")

    seed = cfg['seed']
    if seed == -1:
        seed = np.random.randint(2 ** 32 - 1, dtype=np.int64)
    random.seed(int(seed))
    np.random.seed(int(seed) % (2 ** 32))
    torch.manual_seed(int(seed))
    print(f'running with seed: {seed}.')

    A = _load_image(_first_file(os.path.join(cfg['dataroot'], 'A')), cfg['A_resize'])
    B = _load_image(_first_file(os.path.join(cfg['dataroot'], 'B')), cfg['B_resize'])
    if cfg['direction'] == 'BtoA':
        A, B = B, A
    print("Image sizes %s and %s" % (str((A.shape[2], A.shape[1])), str((B.shape[2], B.shape[1]))))
    feed = DeviceDataFeed(cfg, A, B)

    vit_engine = None if vit_state is not None else _shared_vit_engine(cfg['dino_model_name'], "train_model")
    # generator initialised exactly as define_G(init_type, init_gain): xavier-normal from the torch RNG seeded above
    from .networks import define_G
    netG = define_G(cfg['init_type'], cfg['init_gain'], device=device)
    gen_state = {k: v.detach() for k, v in netG.state_dict().items() if k in netG.engine.table}
    crop_max = max(min(A.shape[1], A.shape[2]), min(B.shape[1], B.shape[2]))   # crops are squares of side <= min(h, w)
    # Extensions beyond the reference's config (BASELINE configs[4]): `dino_global_scales` = list of ViT input sizes at which every
    # loss term is evaluated each step (the reference has the single `dino_global_patch_size`), `fp8` = e4m3 operands for the
    # QKV / fc1 / fc2 projections and the self-similarity Gram matrices.  Absent keys = the reference's behaviour.
    scales = [int(x) for x in (cfg.get('dino_global_scales') or [])]
    fp8 = fp8_mode(cfg)
    crops, entire = (crop_max, crop_max), tuple(A.shape[1:])
    if len(scales) > 1:
        from .engine import MultiScaleEngine
        engine = MultiScaleEngine(cfg, vit_state, gen_state, crops, entire, scales=scales, device=device, n_crops=n_crops, vit_engine=vit_engine, fp8=fp8)
    else:
        if scales:
            cfg['dino_global_patch_size'] = scales[0]
        engine = SpliceEngine(cfg, vit_state, gen_state, crops, entire, device=device, n_crops=n_crops, vit_engine=vit_engine, fp8=fp8)
    del netG

    writer = AsyncResultWriter(cfg['dataroot'])   # PNG encode + disk write happen on a worker thread
    try:
        for epoch in range(1, cfg['n_epochs'] + 1):
            inputs = feed.next()
            log = epoch % cfg['log_images_freq'] == 0
            if log:
                # train.py:70-76 generates the logged image between the loss and backward(): with the weights of epoch - 1
                # updates.  The fused step updates in place, so the image is generated BEFORE it (same weights) ...
                output = engine.generate(feed.get_A())
            engine.step(inputs['A_global'], inputs['B_global'], inputs['A'][0] if 'A' in inputs else None)
            if log:
                engine.book_logged_forward()   # ... and its BatchNorm bookkeeping lands AFTER the step's, as in the reference
                # (out/output.png is overwritten every time: intermediate images may be skipped when they come faster than the writer's interval, the last one never)
                writer.submit(output[0], force=epoch + cfg['log_images_freq'] > cfg['n_epochs'])
                if callback is not None:
                    callback(output[0])
            if progress and (epoch % 50 == 0 or epoch == 1):
                print(f"Epoch {epoch}: loss={engine.losses()['loss']:.4f} lr={cfg['lr']}")
    finally:
        writer.close()
    return engine


class PairBatchFeed:
    """Device data feed of P pairs that share image sizes: per step ONE crop size per side (A / B), as ``Global_crops`` draws it
    (data/transforms.py:21), and one random position per pair; augmentations per pair.  Every pair sees the reference's marginal
    distribution of crops; the sizes are shared so that the P crops stack into one ``[P,3,s,s]`` batch."""

    def __init__(self, cfg, As, Bs):
        if len({tuple(a.shape) for a in As}) != 1 or len({tuple(b.shape) for b in Bs}) != 1:
            raise ValueError("pairs optimised side by side must share the structure-image size and the appearance-image size")
        self.cfg, self.A, self.B = cfg, [a.to(device) for a in As], [b.to(device) for b in Bs]
        self.step = -1

    def get_A(self, pair):
        return self.A[pair][None]

    def _crops(self, imgs, min_cover):
        _, h, w = imgs[0].shape
        size, boxes = augment.global_crop_boxes(h, w, min_cover, len(imgs))
        return torch.stack([im[:, t:t + size, l:l + size] for im, (t, l) in zip(imgs, boxes)]).contiguous()

    def next(self):
        self.step += 1
        aug = bool(self.cfg['use_augmentations'])
        sample = {'step': self.step}
        if self.step % self.cfg['entire_A_every'] == 0:
            sample['A'] = torch.stack(self.A).contiguous()
        sample['A_global'] = self._crops([augment.structure_transforms(a) if aug else a for a in self.A], self.cfg['global_A_crops_min_cover'])
        sample['B_global'] = self._crops([augment.texture_transforms(b) if aug else b for b in self.B], self.cfg['global_B_crops_min_cover'])
        return sample


def train_pairs(dataroots, callback=None, cfg_overrides=None, vit_state=None, progress=True):
    """``train_model`` for P pairs on ONE GPU in the same kernel launches (``MultiPairEngine``): P independent optimisations that
    share only the frozen ViT -- the throughput form of the reference's one-pair-per-process loop (train.py:34-80).  ``dataroots``:
    P directories with ``A/`` and ``B/``; all structure images must share one size and all appearance images one size
    (``A_resize`` / ``B_resize`` apply).  Every pair's generator is initialised as its own ``train_model`` run would
    (the seed is re-applied before each ``define_G``); writes ``<dataroot>/out/output.png`` per pair; ``callback(pair, image)``.
    With deterministic full crops (``use_augmentations: False``, ``min_cover: 1``) the result of every pair is bit-identical to
    its single ``train_model`` run."""
    from .engine import MultiPairEngine
    cfg_path = "conf/default/config.yaml" if os.path.exists("conf/default/config.yaml") else _PKG_CFG
    with open(cfg_path, "r") as f:
        cfg = yaml.safe_load(f)
    cfg.update(cfg_overrides or {})
    if int(cfg['global_A_crops_n_crops']) != 1 or int(cfg['global_B_crops_n_crops']) != 1:
        raise NotImplementedError("train_pairs: several pairs per step take one global crop per image (n_crops is a single-pair option)")
    if len(cfg.get('dino_global_scales') or []) > 1:
        raise NotImplementedError("train_pairs: dino_global_scales with several entries is a single-pair option (train_model / MultiScaleEngine); "
                                  "grouped pairs would silently train single-scale")
    if device.type != 'cuda':
        raise RuntimeError("train_pairs needs an MI355X: the product path has no CPU fallback")
    seed = cfg['seed']
    if seed == -1:
        seed = np.random.randint(2 ** 32 - 1, dtype=np.int64)
    random.seed(int(seed))
    np.random.seed(int(seed) % (2 ** 32))
    print(f'running {len(dataroots)} pairs with seed: {seed}.')
    As, Bs = [], []
    for root in dataroots:
        A = _load_image(_first_file(os.path.join(root, 'A')), cfg['A_resize'])
        B = _load_image(_first_file(os.path.join(root, 'B')), cfg['B_resize'])
        if cfg['direction'] == 'BtoA':
            A, B = B, A
        As.append(A); Bs.append(B)
    feed = PairBatchFeed(cfg, As, Bs)
    vit_engine = None if vit_state is not None else _shared_vit_engine(cfg['dino_model_name'], "train_pairs")
    from .networks import define_G
    gen_states = []
    for _ in dataroots:
        torch.manual_seed(int(seed))     # every pair starts as its own train_model run would
        netG = define_G(cfg['init_type'], cfg['init_gain'], device=device)
        gen_states.append({k: v.detach().clone() for k, v in netG.state_dict().items() if k in netG.engine.table})
        del netG
    torch.manual_seed(int(seed))
    if cfg.get('dino_global_scales'):
        cfg['dino_global_patch_size'] = int(cfg['dino_global_scales'][0])   # one entry: the ViT input size, as in train_model
    A0, B0 = As[0], Bs[0]
    crop_max = max(min(A0.shape[1], A0.shape[2]), min(B0.shape[1], B0.shape[2]))
    engine = MultiPairEngine(cfg, vit_state, gen_states, (crop_max, crop_max), tuple(A0.shape[1:]), device=device, vit_engine=vit_engine, fp8=fp8_mode(cfg))
    writers = [AsyncResultWriter(root) for root in dataroots]
    try:
        for epoch in range(1, cfg['n_epochs'] + 1):
            inputs = feed.next()
            log = epoch % cfg['log_images_freq'] == 0
            outputs = [engine.generate(feed.get_A(p), pair=p) for p in range(len(dataroots))] if log else None
            engine.step(inputs['A_global'], inputs['B_global'], inputs.get('A'))
            if log:
                engine.book_logged_forward()   # BatchNorm bookkeeping of the P logging forwards, after the step's (train.py:70-79)
                for p, out in enumerate(outputs):
                    writers[p].submit(out[0], force=epoch + cfg['log_images_freq'] > cfg['n_epochs'])
                    if callback is not None:
                        callback(p, out[0])
            if progress and (epoch % 50 == 0 or epoch == 1):
                print(f"Epoch {epoch}: loss=" + ", ".join(f"{d['loss']:.4f}" for d in engine.losses()) + f" lr={cfg['lr']}")
    finally:
        for w in writers:
            w.close()
    return engine


if __name__ == '__main__':
    parser = ArgumentParser()
    parser.add_argument("--dataroot", type=str)
    args = parser.parse_args()
    train_model(args.dataroot)
