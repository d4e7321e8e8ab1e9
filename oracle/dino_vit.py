"""fp32 CPU restatement of the DINO VisionTransformer (test infrastructure only).

The reference obtains this model with
``torch.hub.load('facebookresearch/dino:main', model_name)`` (``models/extractor.py:20``)
and relies on these attributes / behaviours (``models/extractor.py:41-49,77,83``):
``model.blocks[i]``, ``.blocks[i].attn``, ``.attn.qkv``, ``.attn.attn_drop``, the
attention module returning a tuple whose ``[0]`` is the token tensor, and
``model(img)`` running the full 12-block forward.

The hub source is third-party, un-vendored and un-pinned (branch ``main``): PARITY
UNPINNED AGAINST THE REFERENCE'S OWN SOURCE for the inside of the ViT.  What stands in for it:
``oracle/pin_vit_hf.py`` / ``tests/test_oracle_hf_pin_cpu.py`` load one seeded DINO-keyed state dict
into this module AND into Hugging Face ``transformers.ViTModel`` (the form the public
``facebook/dino-vit*`` checkpoints are served in; an independent implementation) and the token
tensor behind every block agrees to the last bit; the position-table interpolation (the DINO
recipe's +0.1 nudge) is a different recipe in transformers and stays restated-only.
The architecture below follows the published
DINO ViT: patch-embed Conv2d(3, D, k=p, s=p) -> flatten -> prepend cls token -> add
(bicubically interpolated) position embedding -> depth x [x += proj(MHSA(LN(x)));
x += fc2(GELU_erf(fc1(LN(x))))] -> LN; LayerNorm eps 1e-6; qkv bias on; no dropout.
State-dict key names match the public checkpoints (``cls_token, pos_embed,
patch_embed.proj.*, blocks.{i}.{norm1,norm2,attn.qkv,attn.proj,mlp.fc1,mlp.fc2}.*,
norm.*``) so a real DINO ``.pth`` loads with ``load_state_dict``.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

# name -> (patch, dim, depth, heads); the four names of conf/default/config.yaml:25
DINO_CONFIGS = {
    "dino_vits16": (16, 384, 12, 6),
    "dino_vits8": (8, 384, 12, 6),
    "dino_vitb16": (16, 768, 12, 12),
    "dino_vitb8": (8, 768, 12, 12),
}


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()  # exact erf form
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class Attention(nn.Module):
    def __init__(self, dim, num_heads):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.attn_drop = nn.Dropout(0.0)  # hooked by the reference (extractor.py:44-45)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        attn = (q @ k.transpose(-2, -1)) * self.scale
        attn = self.attn_drop(attn.softmax(dim=-1))
        x = (attn @ v).transpose(1, 2).reshape(B, N, C)
        return self.proj(x), attn  # tuple: extractor.py:77 takes output[0]


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.0):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = Attention(dim, num_heads)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))

    def forward(self, x):
        y, _ = self.attn(self.norm1(x))
        x = x + y
        return x + self.mlp(self.norm2(x))


class PatchEmbed(nn.Module):
    def __init__(self, patch, dim, img_size=224):
        super().__init__()
        self.patch_size = patch
        self.num_patches = (img_size // patch) ** 2
        self.proj = nn.Conv2d(3, dim, kernel_size=patch, stride=patch)

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)


class VisionTransformer(nn.Module):
    def __init__(self, patch=8, dim=768, depth=12, heads=12, img_size=224, mlp_ratio=4.0):
        super().__init__()
        self.patch_size = patch
        self.embed_dim = dim
        self.patch_embed = PatchEmbed(patch, dim, img_size)
        n = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, n + 1, dim))
        self.blocks = nn.ModuleList([Block(dim, heads, mlp_ratio) for _ in range(depth)])
        self.norm = nn.LayerNorm(dim, eps=1e-6)

    def interpolate_pos_encoding(self, x, h_px, w_px):
        """Bicubic resample of the patch position grid when the token grid differs
        from the trained one (K20 in SURVEY.md).  Follows the public DINO recipe,
        including its +0.1 scale-factor nudge."""
        npatch = x.shape[1] - 1
        N = self.pos_embed.shape[1] - 1
        if npatch == N and h_px == w_px:
            return self.pos_embed
        cls_pos = self.pos_embed[:, 0]
        patch_pos = self.pos_embed[:, 1:]
        dim = x.shape[-1]
        g = int(math.sqrt(N))
        h0 = h_px // self.patch_size + 0.1
        w0 = w_px // self.patch_size + 0.1
        patch_pos = F.interpolate(
            patch_pos.reshape(1, g, g, dim).permute(0, 3, 1, 2),
            scale_factor=(h0 / g, w0 / g), mode="bicubic")
        assert int(h0) == patch_pos.shape[-2] and int(w0) == patch_pos.shape[-1]
        patch_pos = patch_pos.permute(0, 2, 3, 1).reshape(1, -1, dim)
        return torch.cat((cls_pos.unsqueeze(0), patch_pos), dim=1)

    def prepare_tokens(self, img):
        B, _, H, W = img.shape
        x = self.patch_embed(img)
        x = torch.cat((self.cls_token.expand(B, -1, -1), x), dim=1)
        return x + self.interpolate_pos_encoding(x, H, W)

    def forward(self, img):
        x = self.prepare_tokens(img)
        for blk in self.blocks:
            x = blk(x)
        return self.norm(x)[:, 0]


def round_weights_bf16(state, dim, head_dim=64):
    """TEST INFRASTRUCTURE: the ViT state dict with every Linear / patch-embedding WEIGHT rounded to bf16 the way the HIP engine packs it
    (splice_amd/csrc/vit_engine.hip pack_linear: one rounding; pack_qkv: the q rows are scaled by head_dim^-1/2 * log2(e) in fp32 FIRST and
    rounded once).  Biases, LayerNorm parameters, class token and position embedding stay fp32 in the engine, and so here.  An fp32 oracle on
    these weights is "the model the engine actually holds": comparing against it separates the engine's input-independent deviation (a fixed
    perturbation of the frozen weights) from the rounding of its activations (tests/test_step_gpu.py, tools/traj_grad_error.py)."""
    import numpy as np
    c = (head_dim ** -0.5) * 1.4426950408889634
    out = {}
    for k, v in state.items():
        t = torch.from_numpy(np.asarray(v)).clone() if not torch.is_tensor(v) else v.clone()
        if k.endswith("attn.qkv.weight"):
            t[:dim] = (t[:dim] * c).bfloat16().float() / c
            t[dim:] = t[dim:].bfloat16().float()
        elif k.endswith(".weight") and t.dim() >= 2:
            t = t.bfloat16().float()
        out[k] = t
    return out


def build_vit(model_name=None, patch=None, dim=None, depth=None, heads=None, img_size=224):
    if model_name is not None:
        patch, dim, depth, heads = DINO_CONFIGS[model_name]
    return VisionTransformer(patch, dim, depth, heads, img_size).eval()


def forward_features(model, img):
    """One forward returning everything the reference hooks capture
    (``models/extractor.py:56-79``): per layer the block output ``[B,T,D]``, the
    attention probabilities ``[B,h,T,T]``, the raw qkv Linear output ``[B,T,3D]`` and
    the attention module's token output ``[B,T,D]``."""
    out = {"block": [], "attn": [], "qkv": [], "patch_imd": []}
    x = model.prepare_tokens(img)
    for blk in model.blocks:
        xn = blk.norm1(x)
        B, N, C = xn.shape
        qkv_raw = blk.attn.qkv(xn)
        h = blk.attn.num_heads
        qkv = qkv_raw.reshape(B, N, 3, h, C // h).permute(2, 0, 3, 1, 4)
        attn = ((qkv[0] @ qkv[1].transpose(-2, -1)) * blk.attn.scale).softmax(dim=-1)
        y = blk.attn.proj((attn @ qkv[2]).transpose(1, 2).reshape(B, N, C))
        x = x + y
        x = x + blk.mlp(blk.norm2(x))
        out["block"].append(x)
        out["attn"].append(attn)
        out["qkv"].append(qkv_raw)
        out["patch_imd"].append(y)
    return out
