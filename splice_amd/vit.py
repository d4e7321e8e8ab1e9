"""Host side of the DINO-ViT engine: weight upload, per-shape contexts, tensor access.

Thin plumbing over the C ABI (``splice_vit_*`` in include/splice_hip.h); PyTorch only
provides device memory and the stream.  The one piece of arithmetic done here is DINO's
``interpolate_pos_encoding`` (bicubic resample of the position grid, K20 in SURVEY.md),
run once per image shape with torch on the device and cached.
"""
import ctypes as C
import math

import torch
import torch.nn.functional as F

from . import _lib
from .synth import DINO_CONFIGS

KIND_BLOCK, KIND_QKV, KIND_ATTN_OUT, KIND_QKV_LAST_F32, KIND_LSE, KIND_TOKENS = 0, 1, 2, 3, 4, 5
KIND_QKV_STORED = 7   # the layer's qkv as the engine holds it: q columns pre-multiplied by VitEngine.qscale (KIND_QKV divides it out of the copy)


def interpolate_pos_encoding(pos_embed, patch, h_px, w_px):
    """pos_embed [1, 1+N, D] (trained grid) -> [T, D] for an h_px x w_px image.  Same recipe
    as the public DINO ViT (incl. its +0.1 scale-factor nudge)."""
    D = pos_embed.shape[-1]
    N = pos_embed.shape[1] - 1
    gh, gw = h_px // patch, w_px // patch
    if gh * gw == N and h_px == w_px:
        return pos_embed[0]
    g = int(math.sqrt(N))
    grid = pos_embed[:, 1:].reshape(1, g, g, D).permute(0, 3, 1, 2)
    grid = F.interpolate(grid, scale_factor=((gh + 0.1) / g, (gw + 0.1) / g), mode="bicubic")
    assert grid.shape[-2] == gh and grid.shape[-1] == gw
    return torch.cat((pos_embed[0, :1], grid.permute(0, 2, 3, 1).reshape(-1, D)), dim=0)


def fp8_mode(v):
    """The ONE meaning of an ``fp8`` argument / config value, everywhere (config key, CLI, engines, contexts):
    falsy -> ``False`` (bf16); ``True`` / ``"gemm"`` -> ``"gemm"``: e4m3 operands for the QKV / fc1 / fc2 forward projections (and,
    in a step engine, the key self-similarity Gram matrices) -- the fastest measured setting; ``"attention"`` / ``"all"`` ->
    ``"attention"``: the attention forward on e4m3 operands as well (BASELINE configs[4] as written; measured 0.5 - 2.8 % slower
    than ``"gemm"``, profiles/r03_fp8_attention_ab.txt)."""
    if isinstance(v, str):
        v = v.strip().lower()
        if v in ("", "false", "0", "off", "no", "none"):
            return False
        if v in ("attention", "all"):
            return "attention"
        if v in ("gemm", "true", "1", "on", "yes"):
            return "gemm"
        raise ValueError(f"fp8 mode {v!r}: expected False, True / 'gemm' or 'attention'")
    return "gemm" if v else False


class VitContext:
    """Activations of one (batch, image-shape) forward; see splice_vit_ctx_create."""

    def __init__(self, engine, B, H, W, need_grad, fp8=None):
        """``fp8`` (see ``fp8_mode``): True / ``"gemm"`` -- this context's QKV / fc1 / fc2 forward projections on the fp8 MFMA;
        ``"attention"`` -- its attention forward too; False -- bf16.  None: the engine's default (``VitEngine.enable_fp8()`` switches it on
        for the contexts the engine hands out itself)."""
        self.engine, self.B, self.H, self.W, self.need_grad = engine, B, H, W, need_grad
        self.fp8 = fp8_mode(getattr(engine, "fp8", False) if fp8 is None else fp8)
        pos = interpolate_pos_encoding(engine.pos_embed, engine.patch, H, W).contiguous().float()
        h = C.c_void_p()
        _lib.check(_lib.lib().splice_vit_ctx_create(engine.handle, B, H, W, _lib.ptr(pos), int(need_grad),
                                                    _lib.current_stream(), C.byref(h)), "vit_ctx_create")
        torch.cuda.current_stream().synchronize()  # pos may be freed after this
        self.handle = h
        self.generation = 0      # forwards run through this context (autograd nodes check theirs is still the resident one)
        t, tld, rows = C.c_int(), C.c_int(), C.c_int()
        _lib.check(_lib.lib().splice_vit_ctx_info(h, C.byref(t), C.byref(tld), C.byref(rows)))
        self.T, self.Tld, self.rows = t.value, tld.value, rows.value
        if self.fp8:
            _lib.check(_lib.lib().splice_vit_ctx_set_fp8(h, 3 if self.fp8 == "attention" else 1), "vit_ctx_set_fp8")

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _lib.lib().splice_vit_ctx_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def forward(self, img, normalize):
        assert img.is_cuda and img.dtype == torch.float32 and img.is_contiguous()
        assert tuple(img.shape) == (self.B, 3, self.H, self.W), (img.shape, (self.B, 3, self.H, self.W))
        self.generation += 1
        _lib.check(_lib.lib().splice_vit_forward(self.handle, _lib.ptr(img), int(normalize), _lib.current_stream()), "vit_forward")

    def read(self, kind, layer):
        """Fresh torch tensor with a copy of an engine tensor (full padded layout)."""
        e = self.engine
        D = e.dim
        if kind in (KIND_BLOCK, KIND_TOKENS):
            t = torch.empty(self.B, self.Tld, D, device=e.device)
        elif kind in (KIND_QKV, KIND_QKV_STORED):
            t = torch.empty(self.B, self.Tld, 3 * D, device=e.device, dtype=torch.bfloat16)
        elif kind == KIND_ATTN_OUT:
            t = torch.empty(self.B, self.Tld, D, device=e.device, dtype=torch.bfloat16)
        elif kind == KIND_QKV_LAST_F32:
            t = torch.empty(self.B, self.Tld, 3 * D, device=e.device)
        elif kind == KIND_LSE:
            t = torch.empty(self.B, e.heads, self.Tld, device=e.device)
        else:
            raise ValueError(kind)
        _lib.check(_lib.lib().splice_vit_read_tensor(self.handle, kind, layer, _lib.ptr(t), t.numel() * t.element_size(),
                                                     _lib.current_stream()), "vit_read_tensor")
        return t

    def backward(self, pass_begin, pass_end, d_block=None, d_qkv=None, d_keys=None, normalize=False):
        """d_*: dict layer -> fp32 tensor in the padded full-batch layout ([B,Tld,D] / [B,Tld,3D])."""
        L = self.engine.depth
        keep = []

        def arr(d):
            if not d:
                return None
            a = (C.c_void_p * L)()
            for l, t in d.items():
                assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
                keep.append(t)
                a[l] = t.data_ptr()
            return a

        d_img = torch.zeros(self.B, 3, self.H, self.W, device=self.engine.device)
        _lib.check(_lib.lib().splice_vit_backward(self.handle, pass_begin, pass_end, arr(d_block), arr(d_qkv), arr(d_keys),
                                                  _lib.ptr(d_img), int(normalize), _lib.current_stream()), "vit_backward")
        return d_img


class VitEngine:
    def __init__(self, model_name=None, patch=None, dim=None, depth=None, heads=None, device="cuda"):
        if model_name is not None:
            patch, dim, depth, heads = DINO_CONFIGS[model_name]
        self.model_name, self.patch, self.dim, self.depth, self.heads = model_name, patch, dim, depth, heads
        self.device = torch.device(device)
        h = C.c_void_p()
        _lib.check(_lib.lib().splice_vit_create(patch, dim, depth, heads, C.byref(h)), "vit_create")
        self.handle = h
        # the engine stores q pre-multiplied by qscale = d^-1/2 log2(e) (always since round 6); kernels fed with stored q take attn_scale
        self.qscale = float(_lib.lib().splice_vit_qscale(h))
        self.attn_scale = math.log(2.0) if self.qscale != 1.0 else (dim // heads) ** -0.5
        self.pos_embed = None
        self._ctx = {}

    def __del__(self):
        try:
            self._ctx.clear()
            if getattr(self, "handle", None):
                _lib.lib().splice_vit_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def load_state_dict(self, sd):
        """sd: DINO checkpoint-style mapping name -> array/tensor (fp32)."""
        L = _lib.lib()
        for name, val in sd.items():
            t = torch.as_tensor(val).to(self.device, torch.float32).contiguous()
            if name == "pos_embed":
                self.pos_embed = t.reshape(1, -1, self.dim).clone()
            _lib.check(L.splice_vit_set_param(self.handle, name.encode(), _lib.ptr(t), t.numel(), _lib.current_stream()),
                       f"vit_set_param({name})")
        torch.cuda.current_stream().synchronize()
        if not L.splice_vit_params_complete(self.handle):
            raise RuntimeError("VitEngine.load_state_dict: state dict is missing DINO ViT entries")
        return self

    def prepare_fp8(self):
        """BASELINE configs[4]: build the e4m3 copies of the QKV / fc1 / fc2 weights (once).  Contexts opt in one by one
        (``VitContext(..., fp8="gemm" | "attention")``), so engines that share this frozen ViT keep their own precision."""
        if not getattr(self, "_fp8_ready", False):
            _lib.check(_lib.lib().splice_vit_enable_fp8(self.handle, _lib.current_stream()), "vit_enable_fp8")
            torch.cuda.current_stream().synchronize()
            self._fp8_ready = True
        return self

    def enable_fp8(self, mode="gemm"):
        """``prepare_fp8`` + make ``mode`` (``"gemm"``: e4m3 QKV / fc1 / fc2 projections; ``"attention"``: the attention forward too)
        the DEFAULT of the contexts this engine hands out through ``context()`` from now on (cached contexts are dropped; this
        includes the contexts an extractor built on this engine creates afterwards -- tolerances of the mode: tests/test_fp8_gpu.py).
        Contexts built with an explicit ``fp8=`` argument are not affected."""
        self.prepare_fp8()
        self._ctx.clear()
        self.fp8 = fp8_mode(mode)
        return self

    def context(self, B, H, W, need_grad=True, fp8=None):
        fp8 = fp8_mode(getattr(self, "fp8", False) if fp8 is None else fp8)
        key = (B, H, W, bool(need_grad), fp8)
        if key not in self._ctx:
            self._ctx[key] = VitContext(self, B, H, W, need_grad, fp8=fp8)
        return self._ctx[key]
