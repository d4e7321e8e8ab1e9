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
* attention probabilities (``get_attn_feature_from_input``) are materialised on request only
  and do not carry gradients (nothing in the reference differentiates through them).
"""
import ctypes as C
import os

import torch

from . import _lib, synth
from .vit import KIND_BLOCK, KIND_LSE, KIND_QKV, KIND_QKV_LAST_F32, VitContext, VitEngine


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
        qkv = torch.stack([vctx.read(KIND_QKV, l)[0, :T].float() for l in range(L - 1)] +
                          [vctx.read(KIND_QKV_LAST_F32, L - 1)[0, :T]])
        ctx.vctx, ctx.extractor, ctx.need_grad = vctx, extractor, need_grad
        if not need_grad:
            extractor._release(vctx)
        return blocks, qkv

    @staticmethod
    def backward(ctx, d_blocks, d_qkv):
        vctx, ext = ctx.vctx, ctx.extractor
        if not ctx.need_grad:
            return None, None, None
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
        ext._release(vctx)
        return d_img, None, None


class VitExtractor:
    BLOCK_KEY = 'block'
    ATTN_KEY = 'attn'
    PATCH_IMD_KEY = 'patch_imd'
    QKV_KEY = 'qkv'
    KEY_LIST = [BLOCK_KEY, ATTN_KEY, PATCH_IMD_KEY, QKV_KEY]

    def __init__(self, model_name, device, state_dict=None, checkpoint=None, synthetic=None, seed=1234, engine=None):
        self.model_name = model_name
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
        eng = self.engine
        img = input_img.to(self.device).detach().contiguous().float()
        _, _, H, W = img.shape
        vctx = self._acquire(H, W, False)
        vctx.forward(img, normalize=False)
        out = []
        L = _lib.lib()
        for l in range(eng.depth):
            qkv = vctx.read(KIND_QKV, l)
            lse = vctx.read(KIND_LSE, l)
            probs = torch.empty(1, eng.heads, vctx.T, vctx.T, device=self.device)
            _lib.check(L.splice_attention_probs(_lib.ptr(qkv), 1, vctx.T, vctx.Tld, eng.dim, eng.heads, 0.125, _lib.ptr(lse),
                                                _lib.ptr(probs), _lib.current_stream()), "attention_probs")
            out.append(probs)
        self._release(vctx)
        return out

    def get_patch_size(self):
        return 8 if "8" in self.model_name else 16

    def get_width_patch_num(self, input_img_shape):
        b, c, h, w = input_img_shape
        patch_size = self.get_patch_size()
        return w // patch_size

    def get_height_patch_num(self, input_img_shape):
        b, c, h, w = input_img_shape
        patch_size = self.get_patch_size()
        return h // patch_size

    def get_patch_num(self, input_img_shape):
        patch_num = 1 + (self.get_height_patch_num(input_img_shape) * self.get_width_patch_num(input_img_shape))
        return patch_num

    def get_head_num(self):
        if "dino" in self.model_name:
            return 6 if "s" in self.model_name else 12
        return 6 if "small" in self.model_name else 12

    def get_embedding_dim(self):
        if "dino" in self.model_name:
            return 384 if "s" in self.model_name else 768
        return 384 if "small" in self.model_name else 768

    def _split(self, qkv, input_img_shape, which):
        patch_num = self.get_patch_num(input_img_shape)
        head_num = self.get_head_num()
        embedding_dim = self.get_embedding_dim()
        return qkv.reshape(patch_num, 3, head_num, embedding_dim // head_num).permute(1, 2, 0, 3)[which]

    def get_queries_from_qkv(self, qkv, input_img_shape):
        return self._split(qkv, input_img_shape, 0)

    def get_keys_from_qkv(self, qkv, input_img_shape):
        return self._split(qkv, input_img_shape, 1)

    def get_values_from_qkv(self, qkv, input_img_shape):
        return self._split(qkv, input_img_shape, 2)

    def get_keys_from_input(self, input_img, layer_num):
        qkv_features = self.get_qkv_feature_from_input(input_img)[layer_num]
        keys = self.get_keys_from_qkv(qkv_features, input_img.shape)
        return keys

    def get_keys_self_sim_from_input(self, input_img, layer_num):
        keys = self.get_keys_from_input(input_img, layer_num=layer_num)
        h, t, d = keys.shape
        concatenated_keys = keys.transpose(0, 1).reshape(t, h * d)
        ssim_map = attn_cosine_sim(concatenated_keys[None, None, ...])
        return ssim_map
