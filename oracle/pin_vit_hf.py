"""Cross-pin of ``oracle/dino_vit.py`` against an independent implementation of the same published architecture
(test infrastructure only; nothing under ``splice_amd/`` imports this).

The reference gets its ViT from ``torch.hub.load('facebookresearch/dino:main', ...)`` (``models/extractor.py:20``): third-party,
un-vendored, absent from ``/root/reference`` and not fetchable here, so the inside of the ViT cannot be pinned to reference
outputs.  What IS in the image is Hugging Face ``transformers`` (5.x), whose ``ViTModel`` is the architecture the public
``facebook/dino-vit{s,b}{8,16}`` checkpoints are served in (converted from the hub weights by transformers'
``convert_dino_to_pytorch.py``: fused ``attn.qkv`` rows split into query / key / value thirds, ``norm1/norm2`` ->
``layernorm_before/after``, ``mlp.fc1/fc2``, ``norm`` -> ``layernorm``, LayerNorm eps 1e-6 is NOT the HF default and has to be
set).  This script loads ONE seeded DINO-keyed state dict (``splice_amd.synth.vit_params``) into both models and compares

  * the token tensor behind every block (HF ``hidden_states``; oracle: forward hooks on ``blocks[i]``),
  * the final-LayerNorm output (all tokens),
  * the interpolated position table at a token grid other than the trained one (HF ``interpolate_pos_encoding=True``
    against the DINO recipe with its +0.1 scale nudge -- the two recipes are NOT the same arithmetic; the difference is
    reported, not asserted to vanish),

and, with ``--write``, freezes the HF outputs in ``tests/golden/vit_hf_pin.npz`` (strided samples, fp32) so that
``tests/test_oracle_hf_pin_cpu.py`` can replay them where ``transformers`` is absent.

    python oracle/pin_vit_hf.py [--write]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import dino_vit  # noqa: E402
from splice_amd import synth  # noqa: E402

# (name, patch, dim, depth, heads, trained image size, evaluated image size)
CASES = [
    ("vits8_64", 8, 384, 12, 6, 64, 64),
    ("vitb16_96", 16, 768, 12, 12, 96, 96),
]
POS_CASE = ("vits8_pos", 8, 384, 2, 6, 64, 96)   # token grid 12 x 12 from a trained 8 x 8 table
SEED_W, SEED_X = 311, 312


def hf_model(patch, dim, depth, heads, img):
    from transformers import ViTConfig, ViTModel
    cfg = ViTConfig(hidden_size=dim, num_hidden_layers=depth, num_attention_heads=heads, intermediate_size=4 * dim, image_size=img,
                    patch_size=patch, layer_norm_eps=1e-6, hidden_act="gelu", qkv_bias=True, hidden_dropout_prob=0.0,
                    attention_probs_dropout_prob=0.0, attn_implementation="eager")
    return ViTModel(cfg, add_pooling_layer=False).eval()


def dino_to_hf(state, dim, depth, hf_keys):
    """The key map of transformers' DINO conversion (fused qkv rows -> q | k | v thirds), onto whichever of the two ViTModel
    key spellings the installed transformers uses."""
    new = "layers.0.attention.q_proj.weight" in hf_keys
    out = {"embeddings.cls_token": state["cls_token"], "embeddings.position_embeddings": state["pos_embed"],
           "embeddings.patch_embeddings.projection.weight": state["patch_embed.proj.weight"],
           "embeddings.patch_embeddings.projection.bias": state["patch_embed.proj.bias"],
           "layernorm.weight": state["norm.weight"], "layernorm.bias": state["norm.bias"]}
    for i in range(depth):
        s = f"blocks.{i}."
        d = f"layers.{i}." if new else f"encoder.layer.{i}."
        names = (("attention.q_proj", "attention.k_proj", "attention.v_proj", "attention.o_proj") if new else
                 ("attention.attention.query", "attention.attention.key", "attention.attention.value", "attention.output.dense"))
        for wb in ("weight", "bias"):
            qkv = state[s + "attn.qkv." + wb]
            for j in range(3):
                out[d + names[j] + "." + wb] = qkv[j * dim:(j + 1) * dim]
            out[d + names[3] + "." + wb] = state[s + "attn.proj." + wb]
            out[d + "layernorm_before." + wb] = state[s + "norm1." + wb]
            out[d + "layernorm_after." + wb] = state[s + "norm2." + wb]
            out[d + ("mlp.fc1." if new else "intermediate.dense.") + wb] = state[s + "mlp.fc1." + wb]
            out[d + ("mlp.fc2." if new else "output.dense.") + wb] = state[s + "mlp.fc2." + wb]
    return out


def run_case(patch, dim, depth, heads, img_train, img_eval, interpolate=False):
    state = {k: torch.from_numpy(v) for k, v in synth.vit_params(SEED_W, patch=patch, dim=dim, depth=depth, img_size=img_train, w_std=0.05).items()}
    x = torch.from_numpy(synth.normal(SEED_X, "pin/img", (2, 3, img_eval, img_eval), 1.0))
    mine = dino_vit.VisionTransformer(patch, dim, depth, heads, img_size=img_train).eval()
    mine.load_state_dict(state)
    hf = hf_model(patch, dim, depth, heads, img_train)
    missing = hf.load_state_dict(dino_to_hf(state, dim, depth, set(hf.state_dict().keys())), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    toks = []
    hooks = [b.register_forward_hook(lambda _m, _i, o: toks.append(o.detach())) for b in mine.blocks]
    with torch.no_grad():
        t = mine.prepare_tokens(x)
        h = t
        for b in mine.blocks:
            h = b(h)
        final_mine = mine.norm(h)
        out = hf(pixel_values=x, output_hidden_states=True, interpolate_pos_encoding=interpolate)
    for hk in hooks:
        hk.remove()
    return dict(x=x, emb_mine=t, emb_hf=out.hidden_states[0], blocks_mine=toks, blocks_hf=list(out.hidden_states[1:]),
                final_mine=final_mine, final_hf=out.last_hidden_state)


def rel(a, b):
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def sample(t, n=4096):
    f = t.reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n].numpy().astype(np.float32)


def main():
    write = "--write" in sys.argv
    gold = {}
    for name, patch, dim, depth, heads, it, ie in CASES:
        r = run_case(patch, dim, depth, heads, it, ie)
        worst = max(rel(a, b) for a, b in zip(r["blocks_mine"], r["blocks_hf"]))
        print(f"{name}: embeddings {rel(r['emb_mine'], r['emb_hf']):.2e}  worst block {worst:.2e}  final LN {rel(r['final_mine'], r['final_hf']):.2e}")
        gold[name + "/final"] = sample(r["final_hf"])
        for i in (0, depth // 2, depth - 1):
            gold[f"{name}/block{i}"] = sample(r["blocks_hf"][i])
    name, patch, dim, depth, heads, it, ie = POS_CASE
    r = run_case(patch, dim, depth, heads, it, ie, interpolate=True)
    print(f"{name}: token embeddings incl. the interpolated position table, DINO recipe vs HF recipe: {rel(r['emb_mine'], r['emb_hf']):.2e} "
          f"(max abs {float((r['emb_mine'] - r['emb_hf']).abs().max()):.2e}); final LN {rel(r['final_mine'], r['final_hf']):.2e}")
    gold[name + "/emb"] = sample(r["emb_hf"])
    if write:
        path = os.path.join(ROOT, "tests", "golden", "vit_hf_pin.npz")
        import transformers
        gold["transformers_version"] = np.array(transformers.__version__)
        np.savez_compressed(path, **gold)
        print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
