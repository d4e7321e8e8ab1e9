"""SpliceEngine: the per-pair optimisation loop of ``train.py:34-80`` on one MI355X.

Owns the frozen DINO-ViT engine, the generator arenas (parameters / gradients / Adam moments,
flat fp32 in ``netG.parameters()`` order) and the fused step handle.  One engine = one image pair
= one GPU = one stream; pairs are independent (no collectives), see ``bench.py --gpus N``.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib, synth
from .generator import GeneratorEngine
from .vit import VitEngine

LOSS_KEYS = ["loss", "loss_global_ssim", "loss_entire_ssim", "loss_entire_cls", "loss_global_cls", "loss_global_id_B"]

DEFAULT_CFG = dict(  # conf/default/config.yaml of the reference
    init_type="xavier", init_gain=0.02,
    lambda_global_cls=10.0, lambda_global_ssim=1.0, lambda_global_identity=1.0,
    entire_A_every=75, lambda_entire_cls=10, lambda_entire_ssim=1.0,
    dino_model_name="dino_vitb8", dino_global_patch_size=224,
    cls_warmup=1, n_epochs=10000, scheduler_policy="none",
    optimizer="adam", optimizer_beta1=0.0, optimizer_beta2=0.99, lr=0.002,
    log_images_freq=10)


def resize_output_size(h, w, size, max_size=480):
    """Output (h, w) of torchvision-0.10 ``Resize(size, max_size)`` (util/losses.py:20): shorter
    edge -> size, aspect kept (long edge truncated), both shrunk if the long edge exceeds max_size;
    unchanged when the shorter edge already equals ``size``."""
    short, long = (w, h) if w <= h else (h, w)
    if short == size:
        return h, w
    new_short, new_long = size, int(size * long / short)
    if max_size is not None and new_long > max_size:
        new_short, new_long = int(max_size * new_short / new_long), max_size
    return (new_long, new_short) if w <= h else (new_short, new_long)


class SpliceEngine:
    def __init__(self, cfg, vit_state, gen_state, crop_hw, entire_hw=None, device="cuda", vit_engine=None):
        """cfg: reference config keys (conf/default/config.yaml); vit_state: DINO state dict;
        gen_state: generator state dict (reference names); crop_hw: (h, w) of the global crops;
        entire_hw: (H, W) of the whole structure image or None to disable the entire branch."""
        self.cfg = dict(DEFAULT_CFG, **cfg)
        c = self.cfg
        if c["optimizer"] != "adam" or c["scheduler_policy"] != "none":
            raise NotImplementedError("SpliceEngine implements the reference's default optimizer 'adam' with scheduler 'none'")
        self.device = torch.device(device)
        self.vit = vit_engine or VitEngine(c["dino_model_name"], device=device).load_state_dict(vit_state)
        self.gen = GeneratorEngine(device=device)
        self.params = self.gen.flatten(gen_state)
        self.grads = torch.zeros_like(self.params)
        self.m = torch.zeros_like(self.params)
        self.v = torch.zeros_like(self.params)
        P = c["dino_global_patch_size"]
        ch, cw = crop_hw
        vh, vw = resize_output_size(ch, cw, P, 480)
        self.crop_hw, self.vit_hw = (ch, cw), (vh, vw)
        self.ctx_g = self.vit.context(4, vh, vw, need_grad=True)
        self.plan_g = self.gen.plan(2, ch, cw, need_grad=True)
        sc = _lib.StepConfig()
        sc.crop_h, sc.crop_w, sc.vit_h, sc.vit_w = ch, cw, vh, vw
        self.ctx_e = self.plan_e = None
        self.entire_hw = entire_hw
        use_entire = entire_hw is not None and (c["lambda_entire_ssim"] > 0 or c["lambda_entire_cls"] > 0)
        if use_entire:
            eh, ew = entire_hw
            evh, evw = resize_output_size(eh, ew, P, 480)
            self.ctx_e = self.vit.context(2, evh, evw, need_grad=True)
            self.plan_e = self.gen.plan(1, eh, ew, need_grad=True)
            sc.ent_h, sc.ent_w, sc.ent_vit_h, sc.ent_vit_w = eh, ew, evh, evw
        sc.lambda_global_cls, sc.lambda_global_ssim = c["lambda_global_cls"], c["lambda_global_ssim"]
        sc.lambda_global_identity = c["lambda_global_identity"]
        sc.lambda_entire_cls, sc.lambda_entire_ssim = c["lambda_entire_cls"], c["lambda_entire_ssim"]
        sc.entire_every, sc.cls_warmup = c["entire_A_every"], c["cls_warmup"]
        sc.lr, sc.beta1, sc.beta2, sc.eps = c["lr"], c["optimizer_beta1"], c["optimizer_beta2"], 1e-8
        h = C.c_void_p()
        _lib.check(_lib.lib().splice_step_create(C.byref(sc), self.ctx_g.handle, self.ctx_e.handle if self.ctx_e else None,
                                                 self.plan_g.handle, self.plan_e.handle if self.plan_e else None, C.byref(h)),
                   "step_create")
        self.handle = h
        self.losses_dev = torch.zeros(8, device=self.device)
        self.step_idx = -1  # data/Dataset.py:57 -- the first step is 0
        self._cur_crops = (ch, cw, ch, cw)
        # two N=1 plans: independent A / B crop sizes (data/Dataset.py:66-67) and, also for equal sizes, two per-image
        # generator chains that run beside each other (faster than the batched N=2 plan)
        # (private plan objects: the shape-keyed plan cache could hand out the entire-image plan of the same size)
        self._split_plans = (GeneratorPlanAlias(self.gen, (ch, cw)), GeneratorPlanAlias(self.gen, (ch, cw)))
        _lib.check(_lib.lib().splice_step_attach_split_plans(self.handle, self._split_plans[0].handle, self._split_plans[1].handle),
                   "step_attach_split_plans")

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _lib.lib().splice_step_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def step(self, A_crop, B_crop, A_entire=None):
        """One optimisation step, asynchronous on the current stream.  Tensors: fp32 CUDA
        ``[3,h,w]`` (or ``[1,3,h,w]``) in [0,1].  Returns the device tensor of 8 losses
        (see LOSS_KEYS); call ``losses()`` to sync and read them."""
        self.step_idx += 1
        for t in (A_crop, B_crop):
            assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        crops = tuple(A_crop.shape[-2:]) + tuple(B_crop.shape[-2:])
        if crops != self._cur_crops:   # per-step random crop sizes (data/transforms.py:21-22)
            _lib.check(_lib.lib().splice_step_set_crops(self.handle, *crops), "step_set_crops")
            self._cur_crops = crops
        if A_entire is not None:
            assert A_entire.is_cuda and A_entire.is_contiguous() and A_entire.numel() == 3 * self.entire_hw[0] * self.entire_hw[1]
        _lib.check(_lib.lib().splice_step_run(self.handle, _lib.ptr(self.params), _lib.ptr(self.grads), _lib.ptr(self.m), _lib.ptr(self.v),
                                              _lib.ptr(A_crop), _lib.ptr(B_crop), _lib.ptr(A_entire), self.step_idx,
                                              _lib.ptr(self.losses_dev), _lib.current_stream()), "step_run")
        return self.losses_dev

    def losses(self):
        """Host dict of the last step's losses, with the reference's keys (inactive terms omitted)."""
        vals = self.losses_dev.cpu().tolist()
        c, s = self.cfg, self.step_idx
        on = s >= c["cls_warmup"]
        ent = self.plan_e is not None and s % c["entire_A_every"] == 0
        active = {"loss": True, "loss_global_ssim": on and c["lambda_global_ssim"] > 0, "loss_entire_ssim": ent and c["lambda_entire_ssim"] > 0,
                  "loss_entire_cls": ent and c["lambda_entire_cls"] > 0, "loss_global_cls": c["lambda_global_cls"] > 0,
                  "loss_global_id_B": on and c["lambda_global_identity"] > 0}
        return {k: vals[i] for i, k in enumerate(LOSS_KEYS) if active[k]}

    def generate(self, img):
        """netG(img) under no_grad (the logging forward of train.py:70-73); img ``[1,3,H,W]``."""
        n, _, h, w = img.shape
        return self.gen.plan(n, h, w, need_grad=False).forward(self.params, img.contiguous())

    def state_dict(self):
        return {k: v.clone() for k, v in self.gen.unflatten(self.params).items()}


def GeneratorPlanAlias(gen, hw):
    """A second, independent N=1 plan of the same maximum size (the plan cache is keyed by shape)."""
    from .generator import GeneratorPlan
    return GeneratorPlan(gen, 1, hw[0], hw[1], True)


def synthetic_engine(cfg, pair_id=0, hw=(224, 224), seed=1234, device="cuda", vit_engine=None, entire=True):
    """Engine + inputs for the BASELINE benchmark configs: seeded synthetic ViT weights,
    xavier generator init and a U[0,1) pair."""
    c = dict(DEFAULT_CFG, **cfg)
    P = c["dino_global_patch_size"]
    vit_state = None
    if vit_engine is None:
        vit_state = synth.vit_params(seed, c["dino_model_name"], img_size=P)
    gen_state = synth.generator_params(seed + 1 + pair_id, c["init_gain"])
    A, B = synth.image_pair(seed, pair_id, hw[0], hw[1])
    eng = SpliceEngine(c, vit_state, gen_state, hw, hw if entire else None, device=device, vit_engine=vit_engine)
    A = torch.from_numpy(A).to(device)
    B = torch.from_numpy(B).to(device)
    return eng, A, B
