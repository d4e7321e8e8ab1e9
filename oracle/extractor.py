"""Restatement of the feature-slicing half of ``models/extractor.py`` (test infra only)."""
import torch

from . import dino_vit


def attn_cosine_sim(x, eps=1e-8):
    """``models/extractor.py:4-9``: x is ``[1,1,T,D]``; cosine similarity of every row
    pair, the clamp applied to the PRODUCT of the two norms.  Returns ``[1,T,T]``."""
    x = x[0]
    n = x.norm(dim=2, keepdim=True)
    return (x @ x.transpose(1, 2)) / torch.clamp(n @ n.transpose(1, 2), min=eps)


def split_qkv(qkv, heads):
    """``models/extractor.py:132-151``: raw qkv ``[1,T,3D]`` -> q,k,v each ``[h,T,d]``
    (only valid for batch 1, as in the reference)."""
    T, D3 = qkv.shape[-2], qkv.shape[-1]
    D = D3 // 3
    r = qkv.reshape(T, 3, heads, D // heads).permute(1, 2, 0, 3)
    return r[0], r[1], r[2]


def keys_from_input(model, img, layer=11):
    """``models/extractor.py:153-156``."""
    feats = dino_vit.forward_features(model, img)
    heads = model.blocks[0].attn.num_heads
    return split_qkv(feats["qkv"][layer], heads)[1]


def keys_self_sim_from_input(model, img, layer=11):
    """``models/extractor.py:158-163``: heads concatenated, all T tokens incl. CLS."""
    k = keys_from_input(model, img, layer)
    h, t, d = k.shape
    return attn_cosine_sim(k.transpose(0, 1).reshape(t, h * d)[None, None])


def cls_from_input(model, img):
    """``util/losses.py:90``: block-11 output row 0, BEFORE the ViT's final LayerNorm."""
    return dino_vit.forward_features(model, img)["block"][-1][0, 0, :]
