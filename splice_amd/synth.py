"""Deterministic synthetic inputs and weights (numpy only; no torch RNG involved).

Counter-based generator: value i of stream ``(seed, tag)`` is
``splitmix64(fnv1a64(tag) ^ (seed * GOLDEN) + i)``, so CPU box, GPU box and every rank
produce bit-identical tensors without shipping them.  Used for
benchmark image pairs ``U[0,1)``, for generator initialisation (xavier-normal, as
``models/networks.py:24-47`` of the reference) and for a DINO-shaped ViT weight set
(trunc-normal-like sigma=0.02, as the public DINO init) when no checkpoint is given.
"""
import math
from collections import OrderedDict

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLDEN = 0x9E3779B97F4A7C15


def _fnv1a64(s: str) -> int:
    h = 0xCBF29CE484222325
    for b in s.encode():
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x):
    with np.errstate(over="ignore"):
        z = x + np.uint64(_GOLDEN)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _bits(seed: int, tag: str, n: int, offset: int = 0):
    key = (_fnv1a64(tag) ^ ((seed * _GOLDEN) & 0xFFFFFFFFFFFFFFFF)) & 0xFFFFFFFFFFFFFFFF
    ctr = np.arange(offset, offset + n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        return _splitmix64(ctr * np.uint64(2) + np.uint64(key))


def uniform(seed: int, tag: str, shape) -> np.ndarray:
    """float32 U[0,1) of ``shape``."""
    n = int(np.prod(shape))
    u = (_bits(seed, tag, n) >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)
    return u.reshape(shape)


def normal(seed: int, tag: str, shape, std=1.0, mean=0.0) -> np.ndarray:
    """float32 N(mean, std) via Box-Muller on two counter streams."""
    n = int(np.prod(shape))
    u1 = ((_bits(seed, tag + "/a", n) >> np.uint64(11)).astype(np.float64) + 1.0) * 2.0 ** -53
    u2 = (_bits(seed, tag + "/b", n) >> np.uint64(11)).astype(np.float64) * 2.0 ** -53
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * math.pi * u2)
    return (mean + std * z).astype(np.float32).reshape(shape)


def image_pair(seed: int, pair_id: int, h: int, w: int):
    """Structure image A and appearance image B, ``[3,h,w]`` float32 in [0,1)."""
    a = uniform(seed, f"pair{pair_id}/A", (3, h, w))
    b = uniform(seed, f"pair{pair_id}/B", (3, h, w))
    return a, b


def smooth_image_pair(seed: int, pair_id: int, h: int, w: int):
    """Low-frequency variant (sum of a few random cosines) -- closer to a photograph
    than white noise; used by the trajectory tests."""
    out = []
    yy, xx = np.meshgrid(np.linspace(0, 1, h, dtype=np.float32),
                         np.linspace(0, 1, w, dtype=np.float32), indexing="ij")
    for tag in ("A", "B"):
        c = uniform(seed, f"pair{pair_id}/{tag}/coef", (3, 6, 4))
        img = np.zeros((3, h, w), np.float32)
        for ch in range(3):
            for k in range(6):
                fx, fy, ph, amp = c[ch, k]
                img[ch] += amp * np.cos(2 * math.pi * (3 * fx * xx + 3 * fy * yy + ph))
        img = (img - img.min()) / (img.max() - img.min() + 1e-6)
        out.append(img.astype(np.float32))
    return out[0], out[1]


# ----------------------------------------------------------------------------- generator
# Parameter table of the reference generator in registration order (state_dict names of
# models/unet/skip.py:46-99 under the nn.Module.add monkey-patch of common.py:6-9).
GEN_DOWN = [16, 32, 64, 128, 128]
GEN_UP = [16, 32, 64, 128, 128]
GEN_SKIP = [4, 4, 4, 4, 4]


def generator_param_specs():
    specs = []

    def conv(name, cout, cin, k):
        specs.append((name + ".weight", (cout, cin, k, k), "conv_w"))
        specs.append((name + ".bias", (cout,), "conv_b"))

    def bn(name, c):
        specs.append((name + ".weight", (c,), "bn_w"))
        specs.append((name + ".bias", (c,), "bn_b"))

    def scale(i, cin):
        p = "1.1.7." * i
        conv(p + "1.0.1.0", GEN_SKIP[i], cin, 1)
        bn(p + "1.0.2", GEN_SKIP[i])
        conv(p + "1.1.1.0", GEN_DOWN[i], cin, 3)
        bn(p + "1.1.2", GEN_DOWN[i])
        conv(p + "1.1.4.0", GEN_DOWN[i], GEN_DOWN[i], 3)
        bn(p + "1.1.5", GEN_DOWN[i])
        if i < 4:
            scale(i + 1, GEN_DOWN[i])
            k = GEN_UP[i + 1]
        else:
            k = GEN_DOWN[i]
        bn(p + "2", GEN_SKIP[i] + k)
        conv(p + "3.0", GEN_UP[i], GEN_SKIP[i] + k, 3)
        bn(p + "4", GEN_UP[i])
        conv(p + "6.0", GEN_UP[i], GEN_UP[i], 1)
        bn(p + "7", GEN_UP[i])

    scale(0, 3)
    conv("9.0", 3, GEN_UP[0], 1)
    return specs


def generator_params(seed: int, init_gain: float = 0.02, perturb_bias: float = 0.0):
    """Seeded xavier-normal(gain) conv weights, zero conv bias, BN gamma ~ N(1, gain),
    BN beta 0 -- the distribution of ``init_weights(..., 'xavier', 0.02)``.
    ``perturb_bias`` > 0 additionally draws biases/betas ~ N(0, perturb_bias) so tests
    exercise the bias paths."""
    out = OrderedDict()
    for name, shape, kind in generator_param_specs():
        if kind == "conv_w":
            cout, cin, k, _ = shape
            std = init_gain * math.sqrt(2.0 / ((cin + cout) * k * k))
            out[name] = normal(seed, "G/" + name, shape, std)
        elif kind == "bn_w":
            out[name] = normal(seed, "G/" + name, shape, init_gain, 1.0)
        elif perturb_bias > 0:
            out[name] = normal(seed, "G/" + name, shape, perturb_bias)
        else:
            out[name] = np.zeros(shape, np.float32)
    return out


# ----------------------------------------------------------------------------- DINO ViT
DINO_CONFIGS = {
    "dino_vits16": (16, 384, 12, 6),
    "dino_vits8": (8, 384, 12, 6),
    "dino_vitb16": (16, 768, 12, 12),
    "dino_vitb8": (8, 768, 12, 12),
}


def vit_param_specs(patch, dim, depth, img_size=224, mlp_ratio=4):
    n = (img_size // patch) ** 2
    hid = dim * mlp_ratio
    specs = [("cls_token", (1, 1, dim), "w"), ("pos_embed", (1, n + 1, dim), "w"),
             ("patch_embed.proj.weight", (dim, 3, patch, patch), "w"),
             ("patch_embed.proj.bias", (dim,), "b")]
    for i in range(depth):
        p = f"blocks.{i}."
        specs += [(p + "norm1.weight", (dim,), "g"), (p + "norm1.bias", (dim,), "b"),
                  (p + "attn.qkv.weight", (3 * dim, dim), "w"), (p + "attn.qkv.bias", (3 * dim,), "b"),
                  (p + "attn.proj.weight", (dim, dim), "w"), (p + "attn.proj.bias", (dim,), "b"),
                  (p + "norm2.weight", (dim,), "g"), (p + "norm2.bias", (dim,), "b"),
                  (p + "mlp.fc1.weight", (hid, dim), "w"), (p + "mlp.fc1.bias", (hid,), "b"),
                  (p + "mlp.fc2.weight", (dim, hid), "w"), (p + "mlp.fc2.bias", (dim,), "b")]
    specs += [("norm.weight", (dim,), "g"), ("norm.bias", (dim,), "b")]
    return specs


def vit_params(seed: int, model_name: str = None, patch=None, dim=None, depth=None,
               img_size=224, w_std=0.02, b_std=0.02, g_std=0.02):
    """DINO-shaped state dict (public checkpoint key names).  Weights N(0, w_std);
    biases N(0, b_std) and LN gains 1+N(0, g_std) instead of DINO's 0 / 1 so every
    bias and gain path is exercised (a trained checkpoint has them non-trivial too)."""
    if model_name is not None:
        patch, dim, depth, _ = DINO_CONFIGS[model_name]
    out = OrderedDict()
    for name, shape, kind in vit_param_specs(patch, dim, depth, img_size):
        tag = "ViT/" + name
        if kind == "w":
            out[name] = normal(seed, tag, shape, w_std)
        elif kind == "b":
            out[name] = normal(seed, tag, shape, b_std)
        else:
            out[name] = normal(seed, tag, shape, g_std, 1.0)
    return out


def vit_params_outlier(seed: int, model_name: str, img_size=224, w_std=0.03, n_outlier=4):
    """A DINO-shaped weight set with the statistics that make trained ViTs hard for bf16: a handful of
    "massive" residual-stream channels (large constant offsets injected by the position table, the patch-embed bias and the
    fc2 biases of the deeper blocks -- in trained DINO these reach tens of sigma), heavy-tailed LayerNorm gains (log-normal,
    with the massive channels squashed as trained models do), and a few large key biases.  Everything else as
    ``vit_params``.  Used by the parity tests to bound the bf16 path's error where the zero-mean N(0, sigma) synthetic set
    cannot."""
    patch, dim, depth, _ = DINO_CONFIGS[model_name]
    out = vit_params(seed, model_name, img_size=img_size, w_std=w_std)
    hot = (np.argsort(uniform(seed, "outlier/channels", (dim,)))[:n_outlier]).astype(np.int64)
    sign = np.where(uniform(seed, "outlier/sign", (n_outlier,)) < 0.5, -1.0, 1.0).astype(np.float32)
    out["pos_embed"][..., hot] += 6.0 * sign
    out["patch_embed.proj.bias"][hot] += 4.0 * sign
    for l in range(depth):
        if l >= depth // 4:
            out[f"blocks.{l}.mlp.fc2.bias"][hot] += 3.0 * sign
        for ln in ("norm1", "norm2"):
            g = out[f"blocks.{l}.{ln}.weight"]
            g *= np.exp(normal(seed, f"outlier/g/{l}/{ln}", (dim,), 0.7)).astype(np.float32)
            g[hot] *= 0.1
        kb = out[f"blocks.{l}.attn.qkv.bias"]
        big = (np.argsort(uniform(seed, f"outlier/kb/{l}", (dim,)))[:3] + dim).astype(np.int64)   # key columns
        kb[big] += 5.0
    return out
