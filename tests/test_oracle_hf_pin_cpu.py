"""The inside of the oracle's DINO ViT (``oracle/dino_vit.py``) against an INDEPENDENT implementation of the same published
architecture: Hugging Face ``transformers.ViTModel``, the form the public ``facebook/dino-vit*`` checkpoints are served in.

The reference's own ViT source (``torch.hub.load('facebookresearch/dino:main', ...)``, models/extractor.py:20) is absent from
``/root/reference`` and cannot be fetched, so this is the strongest pin available offline: one seeded
DINO-keyed state dict goes into both models (key map of transformers' DINO conversion: fused qkv rows -> q | k | v thirds) and
the token tensor behind every block must agree.  ``oracle/pin_vit_hf.py`` is the generator of the frozen vectors.
"""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pin_vit_hf as pin  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden", "vit_hf_pin.npz")
TOL = 2e-6   # fp32 both sides, same operator order (measured: 0 on the generating box; BLAS builds may differ in the last bit)


def _oracle_only(patch, dim, depth, heads, img_train, img_eval):
    from oracle import dino_vit
    from splice_amd import synth
    state = {k: torch.from_numpy(v) for k, v in synth.vit_params(pin.SEED_W, patch=patch, dim=dim, depth=depth, img_size=img_train, w_std=0.05).items()}
    x = torch.from_numpy(synth.normal(pin.SEED_X, "pin/img", (2, 3, img_eval, img_eval), 1.0))
    m = dino_vit.VisionTransformer(patch, dim, depth, heads, img_size=img_train).eval()
    m.load_state_dict(state)
    toks = []
    with torch.no_grad():
        h = m.prepare_tokens(x)
        for b in m.blocks:
            h = b(h)
            toks.append(h)
        return toks, m.norm(h)


@pytest.mark.parametrize("case", pin.CASES, ids=[c[0] for c in pin.CASES])
def test_oracle_vit_matches_frozen_hf_outputs(case):
    name, patch, dim, depth, heads, it, ie = case
    gold = np.load(GOLD)
    toks, final = _oracle_only(patch, dim, depth, heads, it, ie)
    for i in (0, depth // 2, depth - 1):
        want = torch.from_numpy(gold[f"{name}/block{i}"])
        got = torch.from_numpy(pin.sample(toks[i]))
        assert pin.rel(got, want) < TOL, (name, i, pin.rel(got, want))
    assert pin.rel(torch.from_numpy(pin.sample(final)), torch.from_numpy(gold[name + "/final"])) < TOL


def test_oracle_vit_matches_live_hf_model_every_block():
    pytest.importorskip("transformers")
    name, patch, dim, depth, heads, it, ie = pin.CASES[0]
    r = pin.run_case(patch, dim, depth, heads, it, ie)
    assert len(r["blocks_mine"]) == len(r["blocks_hf"]) == depth
    assert pin.rel(r["emb_mine"], r["emb_hf"]) < TOL
    for i, (a, b) in enumerate(zip(r["blocks_mine"], r["blocks_hf"])):
        assert pin.rel(a, b) < TOL, (i, pin.rel(a, b))
    assert pin.rel(r["final_mine"], r["final_hf"]) < TOL
    # and the comparison is not vacuous: a different block's output is far away
    assert pin.rel(r["blocks_mine"][0], r["blocks_hf"][1]) > 1e-2


def test_position_table_recipes_differ_as_documented():
    """DINO's interpolate_pos_encoding (bicubic, scale_factor with the +0.1 nudge; oracle/dino_vit.py) and transformers' (bicubic
    to an explicit size) are different arithmetic: the HF model cannot pin that function, only bound it.  The reference's extractor
    runs the hub model's recipe (models/extractor.py:83 -> VisionTransformer.prepare_tokens), which is what the oracle restates."""
    pytest.importorskip("transformers")
    name, patch, dim, depth, heads, it, ie = pin.POS_CASE
    r = pin.run_case(patch, dim, depth, heads, it, ie, interpolate=True)
    d = pin.rel(r["emb_mine"], r["emb_hf"])
    assert 1e-4 < d < 2e-2, d
