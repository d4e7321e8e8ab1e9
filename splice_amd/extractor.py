"""Drop-in for the reference's ``models/extractor.py`` on the HIP engine.

Same names, argument order and return conventions: ``attn_cosine_sim(x, eps)``, the
``VitExtractor`` class with its KEY constants and 14 public methods
(``models/extractor.py:4-163``).  Differences, all behind the same interface:

* no forward hooks and no torch.hub model: one call into the C ABI (``splice_vit_forward``)
  produces the per-layer tensors the hooks used to capture; feature tensors carry autograd
  w.r.t. the input image through ``splice_vit_backward`` (gradients may be taken from any
  block output / qkv, as with the hooked tensors);
* weights come from a local DINO checkpoint (``checkpoint=`` / ``state_dict=`` / env
  ``SPLICE_DINO_CHECKPOINT``) -- there is no network here; ``synthetic=True`` (or env
  ``SPLICE_SYNTHETIC_WEIGHTS=1``) draws the seeded DINO-shaped weights used by the benchmark;
* attention probabilities (``get_attn_feature_from_input``) are materialised on request only; they carry
  gradients w.r.t. the input image like the hooked tensors (``models/extractor.py:97-103``).
"""
import ctypes as C
import os

import torch

from . import _lib, synth
from .vit import KIND_BLOCK, KIND_LSE, KIND_QKV, KIND_QKV_LAST_F32, KIND_QKV_STORED, VitContext, VitEngine


class _CosineSim(torch.autograd.Function):
    @staticmethod
    def forward(ctx, k, eps):
        T, D = k.shape
        L = _lib.lib()
        ws = torch.empty(L.splice_keys_selfsim_ws_bytes(T, D), device=k.device, dtype=torch.uint8)
        S = torch.empty(T, T, device=k.device)
        _lib.check(L.splice_keys_selfsim_fwd(_lib.ptr(k), k.stride(0), T, D, eps, _lib.ptr(S), _lib.ptr(ws), _lib.current_stream()), "keys_selfsim_fwd")
        ctx.save_for_backward(S, ws)
        ctx.eps, ctx.shape = eps, (T, D)
        return S

    @staticmethod
    def backward(ctx, dS):
        S, ws = ctx.saved_tensors
        T, D = ctx.shape
        dK = torch.empty(T, D, device=S.device)
        dS = dS.contiguous().float()
        _lib.check(_lib.lib().splice_keys_selfsim_bwd(_lib.ptr(dS), _lib.ptr(S), T, D, ctx.eps, _lib.ptr(dK), D, 0, _lib.ptr(ws),
                                                      _lib.current_stream()), "keys_selfsim_bwd")
        return dK, None


def attn_cosine_sim(x, eps=1e-08):
    """``models/extractor.py:4-9``: x ``[1,1,T,D]`` -> cosine self-similarity ``[1,T,T]``."""
    x = x[0]
    if not x.is_cuda:
        raise RuntimeError("attn_cosine_sim: the HIP path needs a CUDA(ROCm) tensor; there is no CPU fallback")
    if x.shape[-1] % 64:
        raise RuntimeError("attn_cosine_sim: feature dim must be a multiple of 64 on the HIP path")
    return _CosineSim.apply(x[0].contiguous().float(), float(eps))[None]


def _release_once(ctx):
    """Hand the node's ViT context back to the extractor's pool exactly once.  A second backward through the same node
    (``retain_graph=True``) still finds the activations in ``ctx.vctx`` -- the context simply stays out of the pool until the
    node dies -- but must not pool it a second time (two later forwards would then share one context)."""
    if not getattr(ctx, "released", False):
        ctx.released = True
        ctx.extractor._release(ctx.vctx)


def _check_resident(ctx):
    if ctx.vctx.generation != ctx.generation:
        raise RuntimeError("VitExtractor: backward through features whose ViT activations were already recycled (a second backward "
                           "with retain_graph=True after another forward of the same image size); recompute the features instead")


class _VitFeatures(torch.autograd.Function):
    """img [1,3,H,W] (normalised) -> (blocks [L,T,D] fp32, qkv [L,T,3D] fp32)."""

    @staticmethod
    def forward(ctx, img, extractor, need_grad):
        eng = extractor.engine
        _, _, H, W = img.shape
        vctx = extractor._acquire(H, W, need_grad)
        vctx.forward(img.contiguous().float(), normalize=False)
        T, L = vctx.T, eng.depth
        blocks = torch.stack([vctx.read(KIND_BLOCK, l)[0, :T] for l in range(L)])
        # the engine stores q pre-multiplied by eng.qscale in bf16: widen FIRST, divide in fp32 -- the hooked 'query' facet then carries the single
        # bf16 rounding of the stored value (KIND_QKV de-scales inside bf16: a second rounding, 2^-9 relative)
        def qkv_f32(l):
            t = vctx.read(KIND_QKV_STORED, l)[0, :T].float()
            if eng.qscale != 1.0:
                t[:, :eng.dim] /= eng.qscale
            return t
        qkv = torch.stack([qkv_f32(l) for l in range(L - 1)] + [vctx.read(KIND_QKV_LAST_F32, L - 1)[0, :T]])
        ctx.vctx, ctx.extractor, ctx.need_grad, ctx.generation = vctx, extractor, need_grad, vctx.generation
        if not need_grad:
            extractor._release(vctx)
        return blocks, qkv

    @staticmethod
    def backward(ctx, d_blocks, d_qkv):
        vctx, ext = ctx.vctx, ctx.extractor
        if not ctx.need_grad:
            return None, None, None
        _check_resident(ctx)
        eng = ext.engine
        T, Tld, D, L = vctx.T, vctx.Tld, eng.dim, eng.depth
        db, dq = {}, {}
        for l in range(L):
            if d_blocks is not None and bool(d_blocks[l].any()):
                t = torch.zeros(1, Tld, D, device=d_blocks.device)
                t[0, :T] = d_blocks[l]
                db[l] = t
            if d_qkv is not None and bool(d_qkv[l].any()):
                t = torch.zeros(1, Tld, 3 * D, device=d_qkv.device)
                t[0, :T] = d_qkv[l]
                dq[l] = t
        d_img = vctx.backward(0, 1, db or None, dq or None, None, normalize=False)
        _release_once(ctx)
        return d_img, None, None


class _VitProbs(torch.autograd.Function):
    """img [1,3,H,W] (normalised) -> attention probabilities [L, heads, T, T] fp32 (softmax(q k^T / sqrt d), the tensor the
    reference hooks behind ``attn_drop``, models/extractor.py:44-45,62-66).  The engine's fused attention never materialises
    them; ``splice_attention_probs`` rebuilds a layer's matrix from its saved q, k and log-sum-exp.  Backward: the softmax
    adjoint dS = P * (dP - rowsum(dP * P)) and the two score products dq = dS k / sqrt d, dk = dS^T q / sqrt d as plain device
    matmuls (this API is off the optimisation path: nothing in train.py reads the probabilities), then the engine's dgrad
    from the layer's qkv down to the image."""

    @staticmethod
    def forward(ctx, img, extractor, need_grad):
        eng = extractor.engine
        _, _, H, W = img.shape
        vctx = extractor._acquire(H, W, need_grad)
        vctx.forward(img.contiguous().float(), normalize=False)
        T = vctx.T
        L = _lib.lib()
        probs = torch.empty(eng.depth, eng.heads, T, T, device=img.device)
        for l in range(eng.depth):
            qkv = vctx.read(KIND_QKV_STORED, l)   # (the q the forward used, bit for bit: the rows must sum to one against ITS log-sum-exp)
            lse = vctx.read(KIND_LSE, l)
            _lib.check(L.splice_attention_probs(_lib.ptr(qkv), 1, T, vctx.Tld, eng.dim, eng.heads, eng.attn_scale, _lib.ptr(lse),
                                                _lib.ptr(probs[l]), _lib.current_stream()), "attention_probs")
        ctx.vctx, ctx.extractor, ctx.need_grad, ctx.generation = vctx, extractor, need_grad, vctx.generation
        if not need_grad:   # (with grad: nothing is saved -- the backward re-forms the matrices it needs from the resident q, k, lse)
            extractor._release(vctx)
        return probs

    @staticmethod
    def backward(ctx, d_probs):
        if not ctx.need_grad:
            return None, None, None
        vctx, ext = ctx.vctx, ctx.extractor
        _check_resident(ctx)
        eng = ext.engine
        T, Tld, D, H = vctx.T, vctx.Tld, eng.dim, eng.heads
        d = D // H
        dq_all = {}
        L = _lib.lib()
        live = d_probs.reshape(eng.depth, -1).any(dim=1).tolist()               # ONE host sync: which layers' probabilities were consumed
        for l in range(eng.depth):
            if not live[l]:
                continue
            dP = d_probs[l]
            P = torch.empty(H, T, T, device=dP.device)                           # re-formed, not saved by the forward ([L, heads, T, T] is 355 MB at T = 785)
            stored = vctx.read(KIND_QKV_STORED, l)
            _lib.check(L.splice_attention_probs(_lib.ptr(stored), 1, T, Tld, D, H, eng.attn_scale, _lib.ptr(vctx.read(KIND_LSE, l)),
                                                _lib.ptr(P), _lib.current_stream()), "attention_probs")
            dS = P * (dP - (dP * P).sum(-1, keepdim=True))                       # [H, T, T]
            qkv = stored[0, :T].float().reshape(T, 3, H, d)
            q, k = qkv[:, 0].transpose(0, 1) / eng.qscale, qkv[:, 1].transpose(0, 1)   # [H, T, d]; seeds below are dL/dq, dL/dk (the engine rescales the q part)
            g = torch.zeros(1, Tld, 3 * D, device=dP.device)
            g[0, :T, :D] = (0.125 * torch.bmm(dS, k)).transpose(0, 1).reshape(T, D)
            g[0, :T, D:2 * D] = (0.125 * torch.bmm(dS.transpose(1, 2), q)).transpose(0, 1).reshape(T, D)
            dq_all[l] = g
        d_img = vctx.backward(0, 1, None, dq_all or None, None, normalize=False) if dq_all else torch.zeros(1, 3, vctx.H, vctx.W, device=d_probs.device)
        _release_once(ctx)
        return d_img, None, None


class VitExtractor:
    BLOCK_KEY = 'block'
    ATTN_KEY = 'attn'
    PATCH_IMD_KEY = 'patch_imd'
    QKV_KEY = 'qkv'
    KEY_LIST = [BLOCK_KEY, ATTN_KEY, PATCH_IMD_KEY, QKV_KEY]

    def __init__(self, model_name, device, state_dict=None, checkpoint=None, synthetic=None, seed=1234, engine=None):
        self.model_name = model_name
        self._arch = _arch_of(model_name)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("VitExtractor (HIP engine) needs a GPU device; there is no CPU fallback")
        if engine is not None:
            self.engine = engine
        else:
            if state_dict is None:
                checkpoint = checkpoint or os.environ.get("SPLICE_DINO_CHECKPOINT")
                if checkpoint:
                    from .checkpoint import load_dino_checkpoint
                    _, state_dict = load_dino_checkpoint(checkpoint, model_name)
                elif synthetic or os.environ.get("SPLICE_SYNTHETIC_WEIGHTS") == "1":
                    state_dict = synth.vit_params(seed, model_name, img_size=224)
                else:
                    raise RuntimeError(
                        "VitExtractor: no DINO weights. The reference downloads them with torch.hub; here pass "
                        "checkpoint=<dino .pth> / state_dict=..., set SPLICE_DINO_CHECKPOINT, or request the seeded "
                        "synthetic weights with synthetic=True / SPLICE_SYNTHETIC_WEIGHTS=1.")
            self.engine = VitEngine(model_name, device=device).load_state_dict(state_dict)
        self.model = self.engine   # the reference exposes .model; here it is the engine handle wrapper
        self._free = {}            # (H, W, need_grad) -> idle contexts (each grad-carrying forward owns one until backward)

    # ---- context pool -------------------------------------------------------------------------
    def _acquire(self, H, W, need_grad):
        pool = self._free.setdefault((H, W, need_grad), [])
        return pool.pop() if pool else VitContext(self.engine, 1, H, W, need_grad)

    def _release(self, vctx):
        self._free.setdefault((vctx.H, vctx.W, bool(vctx.need_grad)), []).append(vctx)

    def _features(self, input_img):
        if input_img.dim() != 4 or input_img.shape[0] != 1:
            raise ValueError("VitExtractor expects a [1,3,H,W] image (the reference's key slicing is batch-1 only)")
        img = input_img.to(self.device)
        return _VitFeatures.apply(img, self, bool(torch.is_grad_enabled() and img.requires_grad))

    # ---- reference API (models/extractor.py:81-163) ------------------------------------------
    def get_feature_from_input(self, input_img):  # List([B, N, D])
        blocks, _ = self._features(input_img)
        return [blocks[l][None] for l in range(blocks.shape[0])]

    def get_qkv_feature_from_input(self, input_img):
        _, qkv = self._features(input_img)
        return [qkv[l][None] for l in range(qkv.shape[0])]

    def get_attn_feature_from_input(self, input_img):
        """The hooked attention probabilities of every layer, List([1, heads, T, T]) (models/extractor.py:97-103).  Like the
        reference's hooked ``attn_drop`` outputs they carry grad with respect to the input image when it requires grad."""
        if input_img.dim() != 4 or input_img.shape[0] != 1:
            raise ValueError("VitExtractor expects a [1,3,H,W] image (the reference's key slicing is batch-1 only)")
        img = input_img.to(self.device)
        probs = _VitProbs.apply(img, self, bool(torch.is_grad_enabled() and img.requires_grad))
        return [probs[l][None] for l in range(probs.shape[0])]

    # ---- shape helpers + q/k/v slicing (models/extractor.py:105-156) -------------------------------------------------
    # The reference derives (patch, heads, width) from sub-strings of the model name on every call; the four DINO names it
    # documents are a table here, any other name goes through the same sub-string rules once (`_arch_of`).
    def get_patch_size(self):
        return self._arch[0]

    def get_head_num(self):
        return self._arch[1]

    def get_embedding_dim(self):
        return self._arch[2]

    def _grid(self, input_img_shape):
        """(patch rows, patch columns) of a ``[b,c,h,w]`` shape."""
        p = self._arch[0]
        return input_img_shape[2] // p, input_img_shape[3] // p

    def get_height_patch_num(self, input_img_shape):
        return self._grid(input_img_shape)[0]

    def get_width_patch_num(self, input_img_shape):
        return self._grid(input_img_shape)[1]

    def get_patch_num(self, input_img_shape):
        rows, cols = self._grid(input_img_shape)
        return rows * cols + 1          # + [CLS]

    def _split(self, qkv, input_img_shape, which):
        """Slice ``which`` (0 q, 1 k, 2 v) of a raw qkv ``[1,T,3D]`` as ``[heads,T,head_dim]`` (batch 1 only)."""
        _, heads, width = self._arch
        tokens = self.get_patch_num(input_img_shape)
        return qkv.reshape(tokens, 3, heads, width // heads)[:, which].transpose(0, 1)

    def get_queries_from_qkv(self, qkv, input_img_shape):
        return self._split(qkv, input_img_shape, 0)

    def get_keys_from_qkv(self, qkv, input_img_shape):
        return self._split(qkv, input_img_shape, 1)

    def get_values_from_qkv(self, qkv, input_img_shape):
        return self._split(qkv, input_img_shape, 2)

    def get_keys_from_input(self, input_img, layer_num):
        raw = self.get_qkv_feature_from_input(input_img)[layer_num]
        return self.get_keys_from_qkv(raw, input_img.shape)

    def get_keys_self_sim_from_input(self, input_img, layer_num):
        """Cosine self-similarity of the layer's keys with the heads laid side by side (models/extractor.py:158-163)."""
        per_head = self.get_keys_from_input(input_img, layer_num=layer_num)          # [heads, T, d]
        tokens = per_head.shape[1]
        token_major = per_head.permute(1, 0, 2).reshape(1, 1, tokens, -1)            # [1, 1, T, heads*d]
        return attn_cosine_sim(token_major)


_DINO_ARCH = {  # model name -> (patch size, heads, embedding width)
    "dino_vits16": (16, 6, 384), "dino_vits8": (8, 6, 384),
    "dino_vitb16": (16, 12, 768), "dino_vitb8": (8, 12, 768),
}


def _arch_of(model_name):
    """Architecture triple of a model name.  Unknown names follow the reference's sub-string conventions
    (models/extractor.py:105-130): an '8' anywhere means patch 8, and the small variant is flagged by an 's' in a dino name
    or by the word 'small' otherwise."""
    if model_name in _DINO_ARCH:
        return _DINO_ARCH[model_name]
    small = ("s" in model_name) if "dino" in model_name else ("small" in model_name)
    return (8 if "8" in model_name else 16,) + ((6, 384) if small else (12, 768))
