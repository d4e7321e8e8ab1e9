"""SpliceEngine / MultiPairEngine: the per-pair optimisation loop of ``train.py:34-80`` on one MI355X.

Owns the frozen DINO-ViT engine, the generator arenas (parameters / gradients / Adam moments,
flat fp32 in ``netG.parameters()`` order, one arena per pair) and the fused step handle.  One engine = one
GPU = one stream; pairs are independent (no collectives): P of them ride one engine's launches
(``MultiPairEngine``), and GPUs run independent engines (``bench.py --gpus N``, ``splice_amd.batch``).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib, synth
from .generator import GeneratorEngine, GeneratorPlan
from .vit import VitContext, VitEngine, fp8_mode

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


class MultiPairEngine:
    """P image pairs optimised side by side on one GPU (``pairs`` = P; P = 1 is the reference's one pair per process).

    The pairs share the frozen ViT and every kernel launch of a step; each has its own generator (parameter / gradient /
    Adam-moment arena ``[P, stride]``), its own BatchNorm statistics and its own loss values.  A pair's trajectory is
    bit-identical whichever batch it rides in (tests/test_multipair_gpu.py).  All pairs of a batch share the image and
    crop sizes."""

    def __init__(self, cfg, vit_state, gen_states, crop_hw, entire_hw=None, device="cuda", vit_engine=None, n_crops=1, fp8=False, top_cls_only=True):
        """cfg: reference config keys (conf/default/config.yaml); vit_state: DINO state dict; gen_states: list of P generator
        state dicts (reference names); crop_hw: (h, w) of the (largest) global crops; entire_hw: (H, W) of the whole
        structure image or None to disable the entire branch.  ``n_crops`` > 1 (one pair only): the reference's
        ``global_{A,B}_crops_n_crops`` -- every step takes ``[n_crops,3,h,w]`` crops of the pair, netG sees them as ONE batch
        (BatchNorm statistics over the crops), every loss term is summed over the crops."""
        self.cfg = dict(DEFAULT_CFG, **cfg)
        c = self.cfg
        # n_crops: int, or (nA, nB) = (global_A_crops_n_crops, global_B_crops_n_crops) -- the reference zips the crop lists
        # (util/losses.py:76,87,98): structure term over the A crops, identity term over the B crops, appearance term over min pairs
        nA, nB = (int(n_crops), int(n_crops)) if isinstance(n_crops, int) else (int(n_crops[0]), int(n_crops[1]))
        self.n_crops_ab = (nA, nB)
        self.n_crops = max(nA, nB)
        if self.n_crops > 1 and len(gen_states) != 1:
            raise ValueError("n_crops > 1 is a property of ONE pair: pass a single generator state")
        if c["optimizer"] != "adam" or c["scheduler_policy"] != "none":
            raise NotImplementedError("the fused step implements the reference's default optimizer 'adam' with scheduler 'none'")
        self.device = torch.device(device)
        self.P = P = len(gen_states)
        self.vit = vit_engine or VitEngine(c["dino_model_name"], device=device).load_state_dict(vit_state)
        # fp8 (vit.fp8_mode; BASELINE configs[4]): True / "gemm" -- the QKV, fc1 and fc2 forward projections (e4m3, block-scaled K = 128
        # MFMA) and the key self-similarity Gram matrices; "attention" -- the attention forward (Q K^T and P V on the fp8 MFMA) too;
        # proj and the whole backward stay bf16 / fp32.  The mode is a property of THIS engine's contexts: another engine sharing
        # the frozen ViT keeps its own.  Tolerances: tests/test_fp8_gpu.py.
        self.fp8 = fp8_mode(fp8)
        if self.fp8:
            self.vit.prepare_fp8()
        self.gen = GeneratorEngine(device=device)
        n = self.gen.numel
        self.stride = n if P == 1 else (n + 63) // 64 * 64
        self.params = torch.zeros(P * self.stride, device=self.device)
        for p_, st in enumerate(gen_states):
            self.params[p_ * self.stride: p_ * self.stride + n] = self.gen.flatten(st)
        self.grads = torch.zeros_like(self.params)
        self.m = torch.zeros_like(self.params)
        self.v = torch.zeros_like(self.params)
        # BatchNorm buffers of every pair's netG (running_mean 0 / running_var 1 at construction, as nn.BatchNorm2d)
        self.running = torch.zeros(P, self.gen.buffer_numel, device=self.device)
        for name, (off, cnt) in self.gen.buffer_table.items():
            if name.endswith("running_var"):
                self.running[:, off:off + cnt] = 1.0
        self.generator_calls = [0] * len(gen_states)   # per pair: every BatchNorm's num_batches_tracked (one per netG call)
        Pz = c["dino_global_patch_size"]
        ch, cw = crop_hw
        vh, vw = resize_output_size(ch, cw, Pz, 480)
        self.crop_hw, self.vit_hw = (ch, cw), (vh, vw)
        slots = self.slots = self.n_crops if self.n_crops > 1 else P    # images per generator plan / per ViT pass group (maximum of the two sides)
        sa, sb = (nA, nB) if self.n_crops > 1 else (P, P)
        self.slots_ab = (sa, sb)
        batch = self.n_crops > 1
        # A step engine's contexts are PRIVATE, never taken from the shape-keyed cache of ``VitEngine.context()``: with an identity
        # Resize the staged inputs, the generator outputs and their gradients live in the context's image slots, and the
        # [CLS]-only mode is a property of the context -- two engines (or an engine and the extractor API) sharing one would
        # overwrite each other's images.
        self.ctx_g = VitContext(self.vit, 2 * (sa + sb), vh, vw, True, fp8=self.fp8)
        arena_stride = self.stride if P > 1 else 0
        # private plan objects (the shape-keyed plan cache could hand out one plan twice)
        self.plan_a = GeneratorPlan(self.gen, sa, ch, cw, True, arena_stride, batch_stats=batch and sa > 1)
        self.plan_b = GeneratorPlan(self.gen, sb, ch, cw, True, arena_stride, batch_stats=batch and sb > 1)
        sc = _lib.StepConfig()
        sc.crop_h, sc.crop_w, sc.vit_h, sc.vit_w = ch, cw, vh, vw
        sc.pairs, sc.arena_stride, sc.fp8_selfsim = P, arena_stride, int(bool(self.fp8))
        sc.n_crops, sc.n_crops_b = (nA, nB) if self.n_crops > 1 else (1, 0)
        # behind the last QKV projection only the [CLS] rows go on (all the losses read of the top block besides its keys):
        # +2.7 % / +5.2 % pair-steps/s at 4 / 8 pairs per GPU, neutral at one pair; top_cls_only=False
        # computes the whole top block as the reference does
        sc.top_cls_only = int(bool(top_cls_only))
        self.ctx_e = self.plan_e = None
        self.entire_hw = entire_hw
        use_entire = entire_hw is not None and (c["lambda_entire_ssim"] > 0 or c["lambda_entire_cls"] > 0)
        if use_entire:
            eh, ew = entire_hw
            evh, evw = resize_output_size(eh, ew, Pz, 480)
            self.ctx_e = VitContext(self.vit, 2 * P, evh, evw, True, fp8=self.fp8)   # (P = 1 in crops mode)
            self.plan_e = GeneratorPlan(self.gen, P, eh, ew, True, arena_stride)
            sc.ent_h, sc.ent_w, sc.ent_vit_h, sc.ent_vit_w = eh, ew, evh, evw
        sc.lambda_global_cls, sc.lambda_global_ssim = c["lambda_global_cls"], c["lambda_global_ssim"]
        sc.lambda_global_identity = c["lambda_global_identity"]
        sc.lambda_entire_cls, sc.lambda_entire_ssim = c["lambda_entire_cls"], c["lambda_entire_ssim"]
        sc.entire_every, sc.cls_warmup = c["entire_A_every"], c["cls_warmup"]
        sc.lr, sc.beta1, sc.beta2, sc.eps = c["lr"], c["optimizer_beta1"], c["optimizer_beta2"], 1e-8
        h = C.c_void_p()
        _lib.check(_lib.lib().splice_step_create(C.byref(sc), self.ctx_g.handle, self.ctx_e.handle if self.ctx_e else None,
                                                 self.plan_a.handle, self.plan_b.handle, self.plan_e.handle if self.plan_e else None, C.byref(h)),
                   "step_create")
        self.handle = h
        _lib.check(_lib.lib().splice_step_set_running_stats(self.handle, _lib.ptr(self.running), self.running.stride(0)), "step_set_running_stats")
        self.losses_dev = torch.zeros(P, 8, device=self.device)
        self.step_idx = -1  # data/Dataset.py:57 -- the first step is 0
        self._cur_crops = (ch, cw, ch, cw)
        self._log_plans = {}

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _lib.lib().splice_step_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def step(self, A_crop, B_crop, A_entire=None, _repeat=False):
        """One optimisation step of every pair, asynchronous on the current stream.  Tensors: fp32 CUDA ``[P,3,h,w]`` in
        [0,1] (``[3,h,w]`` accepted for P = 1).  Returns the device tensor ``[P,8]`` of losses (see LOSS_KEYS).
        ``_repeat``: another phase of the step just run (``splice_step_set_phases``): no bookkeeping."""
        if not _repeat:
            self.step_idx += 1
        for t, n in ((A_crop, self.slots_ab[0]), (B_crop, self.slots_ab[1])):
            assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
            assert t.numel() == n * 3 * t.shape[-2] * t.shape[-1], (tuple(t.shape), n)
        crops = tuple(A_crop.shape[-2:]) + tuple(B_crop.shape[-2:])
        if crops != self._cur_crops:   # per-step random crop sizes (data/transforms.py:21-22)
            _lib.check(_lib.lib().splice_step_set_crops(self.handle, *crops), "step_set_crops")
            self._cur_crops = crops
        entire = self.plan_e is not None and self.step_idx % self.cfg["entire_A_every"] == 0
        if A_entire is not None:
            assert A_entire.is_cuda and A_entire.is_contiguous() and A_entire.numel() == self.P * 3 * self.entire_hw[0] * self.entire_hw[1]
        _lib.check(_lib.lib().splice_step_run(self.handle, _lib.ptr(self.params), _lib.ptr(self.grads), _lib.ptr(self.m), _lib.ptr(self.v),
                                              _lib.ptr(A_crop), _lib.ptr(B_crop), _lib.ptr(A_entire), self.step_idx,
                                              _lib.ptr(self.losses_dev), _lib.current_stream()), "step_run")
        if not _repeat:
            self.generator_calls = [n + (3 if entire else 2) for n in self.generator_calls]   # models/model.py:15-23: G(A_global) [, G(A)], G(B_global)
        return self.losses_dev

    def losses(self, pair=None):
        """Host dict(s) of the last step's losses with the reference's keys (inactive terms omitted): one dict for ``pair``,
        a list over the pairs for ``pair=None``."""
        rows = self.losses_dev.cpu().tolist()
        c, s = self.cfg, self.step_idx
        on = s >= c["cls_warmup"]
        ent = self.plan_e is not None and s % c["entire_A_every"] == 0
        active = {"loss": True, "loss_global_ssim": on and c["lambda_global_ssim"] > 0, "loss_entire_ssim": ent and c["lambda_entire_ssim"] > 0,
                  "loss_entire_cls": ent and c["lambda_entire_cls"] > 0, "loss_global_cls": c["lambda_global_cls"] > 0,
                  "loss_global_id_B": on and c["lambda_global_identity"] > 0}
        dicts = [{k: vals[i] for i, k in enumerate(LOSS_KEYS) if active[k]} for vals in rows]
        return dicts if pair is None else dicts[pair]

    def pair_params(self, pair=0):
        """View of one pair's parameter arena (``numel`` floats)."""
        return self.params[pair * self.stride: pair * self.stride + self.gen.numel]

    def pair_grads(self, pair=0):
        return self.grads[pair * self.stride: pair * self.stride + self.gen.numel]

    def generate(self, img, pair=0, track_running_stats=False):
        """netG_pair(img) under no_grad (the logging forward of train.py:70-73); img ``[n,3,H,W]``.  The reference's net
        is in train mode there too, so the call also moves the BatchNorm running statistics: pass
        ``track_running_stats=True`` to book that (train_model does, after the step whose forwards precede it)."""
        n, _, h, w = img.shape
        key = (pair, n, h, w)   # one plan per pair: a plan holds the BatchNorm statistics of its last forward until they are booked
        if key not in self._log_plans:
            self._log_plans[key] = GeneratorPlan(self.gen, n, h, w, False, batch_stats=n > 1)   # ONE netG call on n images: batch statistics, as nn.BatchNorm2d
        plan = self._log_plans[key]
        out = plan.forward(self.pair_params(pair), img.contiguous())
        if track_running_stats:
            self.book_running_stats(plan, pair)
        else:
            self._logged = [(q, k) for q, k in getattr(self, "_logged", []) if k != pair] + [(plan, pair)]
        return out

    def book_logged_forward(self):
        """Book the BatchNorm statistics of the last ``generate()`` call NOW: the reference's logging forward sits between
        the step's generator calls and the optimizer update (train.py:70-79), so ``train_model`` generates with the
        pre-update weights BEFORE the fused step and books the statistics AFTER it -- the reference's buffer order."""
        for plan, pair in getattr(self, "_logged", []):   # (every pair generated since the last booking, in call order)
            self.book_running_stats(plan, pair)
        self._logged = []

    def book_running_stats(self, plan, pair=0):
        plans = (C.c_void_p * 1)(plan.handle)
        _lib.check(_lib.lib().splice_gen_running_stats_update(plans, 1, _lib.ptr(self.running[pair]), 0, 0.1, _lib.current_stream()), "running_stats_update")
        self.generator_calls[pair] += 1

    def state_dict(self, pair=0):
        """``netG.state_dict()`` of one pair: parameters, BatchNorm running statistics and ``num_batches_tracked``."""
        out = {k: v.clone() for k, v in self.gen.unflatten(self.pair_params(pair)).items()}
        for name, (off, cnt) in self.gen.buffer_table.items():
            out[name] = self.running[pair, off:off + cnt].clone()
            if name.endswith("running_var"):
                out[name[:-len("running_var")] + "num_batches_tracked"] = torch.tensor(self.generator_calls[pair], dtype=torch.long, device=self.device)
        return out


class SpliceEngine(MultiPairEngine):
    """The per-pair optimisation loop of ``train.py:34-80`` for ONE pair (P = 1): the reference's unit of work."""

    def __init__(self, cfg, vit_state, gen_state, crop_hw, entire_hw=None, device="cuda", vit_engine=None, n_crops=1, fp8=False):
        super().__init__(cfg, vit_state, [gen_state], crop_hw, entire_hw, device=device, vit_engine=vit_engine, n_crops=n_crops, fp8=fp8)

    def losses(self):
        return super().losses(0)


class MultiScaleEngine:
    """One pair, every loss term evaluated at SEVERAL ViT input scales (BASELINE configs[4]: 224 / 320 / 448): the same global
    crops are resized to each ``dino_global_patch_size`` in ``scales`` and the reference loss (util/losses.py:46-72) of every
    scale is summed; one Adam update per step on the summed gradient.  An extension beyond the reference (it has one
    scale).  The generator runs ONCE per step: the first scale's engine (the leader) does the generator forward and its own
    ViT part, the other scales (followers, ``splice_step_set_phases``) run only their ViT part on the leader's images and add
    their image gradients to the leader's, then the leader backpropagates the summed image gradient through the generator
    (linear in it, so this equals the sum of the per-scale backpropagations) and the fused Adam launch follows."""

    def __init__(self, cfg, vit_state, gen_state, crop_hw, entire_hw=None, scales=(224, 320, 448), device="cuda", vit_engine=None, n_crops=1,
                 fp8=False):
        self.cfg = dict(DEFAULT_CFG, **cfg)
        self.scales = tuple(scales)
        self.engines = []
        for k, sz in enumerate(self.scales):
            e = SpliceEngine(dict(self.cfg, dino_global_patch_size=sz), vit_state if k == 0 else None, gen_state, crop_hw, entire_hw, device=device,
                             vit_engine=vit_engine if k == 0 else self.engines[0].vit, n_crops=n_crops, fp8=fp8)
            _lib.check(_lib.lib().splice_step_set_mode(e.handle, 1, 0), "step_set_mode")
            if k > 0:   # one parameter set: every scale sees the arenas of the first engine; netG bookkeeping once
                e.params, e.grads, e.m, e.v = self.engines[0].params, self.engines[0].grads, self.engines[0].m, self.engines[0].v
                _lib.check(_lib.lib().splice_step_set_running_stats(e.handle, None, 0), "step_set_running_stats")
                _lib.check(_lib.lib().splice_step_set_phases(e.handle, 2, self.engines[0].handle), "step_set_phases")
            self.engines.append(e)
        self.vit, self.gen = self.engines[0].vit, self.engines[0].gen
        self.params, self.grads = self.engines[0].params, self.engines[0].grads
        self.step_idx = -1

    def step(self, A_crop, B_crop, A_entire=None):
        from .generator import adam_step
        self.step_idx += 1
        e0, c, L = self.engines[0], self.cfg, _lib.lib()
        _lib.check(L.splice_step_set_phases(e0.handle, 1 | 2, None), "step_set_phases")
        e0.step(A_crop, B_crop, A_entire)                    # G forward, the leader's ViT part
        for e in self.engines[1:]:
            e.step(A_crop, B_crop, A_entire)                 # the other scales' ViT parts: d(images) += ...
        _lib.check(L.splice_step_set_phases(e0.handle, 4, None), "step_set_phases")
        e0.step(A_crop, B_crop, A_entire, _repeat=True)      # G backward of the summed image gradient
        adam_step(e0.params, e0.grads, e0.m, e0.v, c["lr"], c["optimizer_beta1"], c["optimizer_beta2"], 1e-8, self.step_idx + 1)

    def losses(self):
        """Per-scale loss dicts and their sum: ``{"loss": total, "scales": {224: {...}, ...}}``."""
        per = {sz: e.losses() for sz, e in zip(self.scales, self.engines)}
        return {"loss": sum(d["loss"] for d in per.values()), "scales": per}

    def generate(self, img, pair=0, track_running_stats=False):
        return self.engines[0].generate(img, pair, track_running_stats)

    def book_logged_forward(self):
        self.engines[0].book_logged_forward()

    def state_dict(self, pair=0):
        return self.engines[0].state_dict(pair)


def synthetic_engine(cfg, pair_id=0, hw=(224, 224), seed=1234, device="cuda", vit_engine=None, entire=True, pairs=1, fp8=False, top_cls_only=True, crop_hw=None):
    """Engine + inputs for the BASELINE benchmark configs: seeded synthetic ViT weights, xavier generator init and U[0,1)
    pairs.  ``pairs`` > 1: pairs ``pair_id .. pair_id + pairs - 1`` side by side on one engine
    (inputs ``[P,3,h,w]``); ``pairs == 1``: the single-pair ``SpliceEngine`` with ``[3,h,w]`` inputs.
    ``crop_hw``: the global crops are the top-left ``crop_hw`` window of the ``hw`` images (the reference's default shape: 900 x 900 crops of
    900 x 1200 images, every crop resized to ``dino_global_patch_size``); the return value then carries the entire image as a fourth entry."""
    if crop_hw is not None:
        if pairs != 1:
            raise ValueError("crop_hw: one pair per engine")
        c = dict(DEFAULT_CFG, **cfg)
        vit_state = None if vit_engine is not None else synth.vit_params(seed, c["dino_model_name"], img_size=c["dino_global_patch_size"])
        a, b = synth.image_pair(seed, pair_id, hw[0], hw[1])
        eng = SpliceEngine(c, vit_state, synth.generator_params(seed + 1 + pair_id, c["init_gain"]), crop_hw, hw if entire else None, device=device, vit_engine=vit_engine, fp8=fp8)
        A, B = torch.from_numpy(a).to(device), torch.from_numpy(b).to(device)
        return eng, A[:, :crop_hw[0], :crop_hw[1]].contiguous(), B[:, :crop_hw[0], :crop_hw[1]].contiguous(), A
    c = dict(DEFAULT_CFG, **cfg)
    P = c["dino_global_patch_size"]
    vit_state = None
    if vit_engine is None:
        vit_state = synth.vit_params(seed, c["dino_model_name"], img_size=P)
    ids = [pair_id + k for k in range(pairs)]
    gen_states = [synth.generator_params(seed + 1 + i, c["init_gain"]) for i in ids]
    imgs = [synth.image_pair(seed, i, hw[0], hw[1]) for i in ids]
    if pairs == 1:
        eng = SpliceEngine(c, vit_state, gen_states[0], hw, hw if entire else None, device=device, vit_engine=vit_engine, fp8=fp8)
        return eng, torch.from_numpy(imgs[0][0]).to(device), torch.from_numpy(imgs[0][1]).to(device)
    eng = MultiPairEngine(c, vit_state, gen_states, hw, hw if entire else None, device=device, vit_engine=vit_engine, fp8=fp8, top_cls_only=top_cls_only)
    A = torch.from_numpy(np.stack([a for a, _ in imgs])).to(device)
    B = torch.from_numpy(np.stack([b for _, b in imgs])).to(device)
    return eng, A, B
