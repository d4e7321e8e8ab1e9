"""GPU parity of the fused optimisation step (C ABI splice_step_*) against the loss trajectories
recorded from the REFERENCE loop (Model + LossG + Adam, oracle/make_golden.py) and the fp32 oracle.

Tolerances (bf16 ViT, fp32 generator/losses/Adam): every entry of the loss dict within 3e-2
relative at steps 0-2 (identical parameters on both sides up to one or two updates).  After that
the trajectory is CHAOTIC in the optimiser itself: Adam with beta1=0 takes ~lr-sized steps along
sign(g), so any gradient perturbation re-routes it.  Measured with the fp32 CPU oracle (DESIGN.md
"trajectory sensitivity"): 2 % multiplicative gradient noise moves the loss at step 72 of this very
fixture from 99.8 to 58..84, i.e. pointwise agreement beyond ~10 steps is not a property the
reference itself has.  So: 6-step window means within 25 % up to step 30, and the optimisation must
end at least as low as 1.25x the reference's level.
"""
import os

import numpy as np
import pytest
import torch

from splice_amd import synth
from splice_amd.engine import LOSS_KEYS, SpliceEngine

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _engine(cfg_over, A, B, gen_seed, img_size):
    cfg = dict(dino_model_name="dino_vits8", dino_global_patch_size=img_size, **cfg_over)
    vit_state = synth.vit_params(7, "dino_vits8", img_size=img_size, w_std=0.05)
    gen_state = synth.generator_params(gen_seed, 0.02)
    return SpliceEngine(cfg, vit_state, gen_state, A.shape[-2:], A.shape[-2:])


def _run(eng, A, B, n):
    rows = []
    A, B = torch.from_numpy(A).to(DEV), torch.from_numpy(B).to(DEV)
    for _ in range(n):
        eng.step(A, B, A)
        rows.append(eng.losses())
    return rows


def _check(rows, gl, keys, lo, hi, rtol):
    worst = 0.0
    for i in range(lo, hi):
        for j, k in enumerate(keys):
            if np.isnan(gl[i, j]):
                assert k not in rows[i], (i, k)
            else:
                rel = abs(rows[i][k] - gl[i, j]) / abs(gl[i, j])
                worst = max(worst, rel)
                assert rel < rtol, (i, k, rows[i][k], gl[i, j])
    print(f"    steps {lo}..{hi - 1}: worst rel loss deviation {worst:.3e}")


def test_trajectory_a_identity_resize(golden_dir):
    g = np.load(os.path.join(golden_dir, "steps.npz"))
    keys = [str(k) for k in g["loss_keys"]]
    assert keys == LOSS_KEYS
    A, B = synth.smooth_image_pair(32, 0, 64, 64)
    eng = _engine({}, A, B, 31, 64)
    rows = _run(eng, A, B, 78)
    _check(rows, g["a/losses"], keys, 0, 3, 3e-2)
    mine = np.array([r["loss"] for r in rows])
    ref = g["a/losses"][:, 0]
    for lo in range(1, 31, 6):     # window means (step 0 excluded: 3x larger, checked exactly above)
        a, b = mine[lo:lo + 6].mean(), ref[lo:lo + 6].mean()
        print(f"    steps {lo}..{lo + 5}: mean loss {a:.1f} vs reference {b:.1f}")
        assert abs(a - b) / b < 0.25, (lo, a, b)
    tail_mine, tail_ref = np.sort(mine[60:75])[:5].mean(), np.sort(ref[60:75])[:5].mean()
    print(f"    level reached (steps 60..74): {tail_mine:.1f} vs reference {tail_ref:.1f}")
    assert tail_mine < 1.25 * tail_ref
    assert np.isfinite(mine).all()
    out = eng.generate(torch.from_numpy(A)[None].to(DEV)).cpu().numpy()
    refimg = g["a/final_out"]
    mse = float(((out - refimg) ** 2).mean())
    psnr = 10 * np.log10(1.0 / max(mse, 1e-12))
    print(f"    final image PSNR(HIP engine vs reference CPU fp32) after 78 steps: {psnr:.1f} dB (reported, chaotic)")
    assert out.shape == refimg.shape and np.isfinite(out).all() and 0.0 <= out.min() and out.max() <= 1.0


def test_trajectory_b_resize_nonsquare(golden_dir):
    """48x80 pair: Resize 48->64 (non-identity, differentiable) and an 8x13 token grid (interpolated
    position embedding)."""
    g = np.load(os.path.join(golden_dir, "steps.npz"))
    keys = [str(k) for k in g["loss_keys"]]
    A, B = synth.smooth_image_pair(34, 1, 48, 80)
    eng = _engine({}, A, B, 33, 64)
    assert eng.vit_hw == (64, 106)
    rows = _run(eng, A, B, 4)
    _check(rows, g["b/losses"], keys, 0, 2, 3e-2)
    _check(rows, g["b/losses"], keys, 2, 4, 2.5e-1)


def test_step_gradients_vs_oracle_teacher_forced():
    """Steps 0..4 (all lambda regimes: step 0 = entire + cls, steps >= 1 = ssim + cls + id) with the
    engine's parameters re-synchronised to the oracle's before every step, so both sides evaluate
    the SAME point: every loss entry within 3e-2 and the whole generator gradient within 5e-2
    (relative L2, bf16 ViT) of the fp32 oracle's autograd through the reference-shaped graph
    (6 ViT forwards / 3 backwards per step).  Without the re-sync the first Adam steps (lr 2e-3 on
    weights initialised at ~1e-3) already make the two parameter sets differ in a few % of the
    signs, and gradients at those two points are no longer comparable (cos ~0.3, measured)."""
    from oracle import dino_vit
    from oracle.step import SpliceOracle
    A, B = synth.smooth_image_pair(40, 2, 64, 64)
    cfg = dict(dino_model_name="dino_vits8", dino_global_patch_size=64)
    vit_state = synth.vit_params(7, "dino_vits8", img_size=64, w_std=0.05)
    gen_state = synth.generator_params(41, 0.02)
    eng = SpliceEngine(cfg, vit_state, gen_state, (64, 64), (64, 64))
    m = dino_vit.VisionTransformer(8, 384, 12, 6, img_size=64).eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in vit_state.items()})
    orc = SpliceOracle(m, {k: torch.from_numpy(v) for k, v in gen_state.items()}, cfg)
    At, Bt = torch.from_numpy(A), torch.from_numpy(B)
    Ad, Bd = At.to(DEV), Bt.to(DEV)
    for step in range(5):
        eng.params.copy_(eng.gen.flatten({k: v.detach() for k, v in orc.params.items()}))
        lo, _, og = orc.step(At[None], Bt[None], At[None])
        eng.step(Ad, Bd, Ad)
        le = eng.losses()
        assert set(le) == set(lo), (le.keys(), lo.keys())
        for k in lo:
            assert abs(le[k] - lo[k]) / abs(lo[k]) < 3e-2, (step, k, le[k], lo[k])
        got = eng.gen.unflatten(eng.grads)
        num = den = 0.0
        for (name, gt), go in zip(got.items(), og):
            if name.endswith("0.bias") and name != "9.0.bias":
                continue  # zero-gradient conv biases (feed a BatchNorm)
            d = (gt.cpu().double() - go.reshape(-1).double()).norm().item()
            num, den = num + d * d, den + go.double().norm().item() ** 2
        rel = (num / den) ** 0.5
        print(f"    step {step}: loss {le['loss']:.3f} vs {lo['loss']:.3f}; generator-gradient rel err vs oracle {rel:.3e}")
        assert rel < 5e-2, rel


def test_streams_and_graph_do_not_change_results():
    """The two-stream hipGraph replay, the two-stream eager path and the single-stream eager path launch the same
    deterministic kernels: parameters after 5 steps (incl. the entire-image branch at step 0, the warm-up switch at
    step 2 and unequal crops at step 3) must agree BIT FOR BIT."""
    from splice_amd import _lib
    A, B = synth.smooth_image_pair(77, 0, 64, 64)
    A2 = np.ascontiguousarray(A[:, :60, :60])
    outs = []
    for graph, overlap in ((1, 1), (0, 1), (0, 0)):
        eng = _engine(dict(cls_warmup=2, entire_A_every=4), A, B, gen_seed=5, img_size=64)
        _lib.check(_lib.lib().splice_step_use_graph(eng.handle, graph))
        _lib.check(_lib.lib().splice_step_use_overlap(eng.handle, overlap))
        At, Bt, A2t = (torch.from_numpy(x).to(DEV) for x in (A, B, A2))
        for i in range(5):
            eng.step(A2t if i == 3 else At, Bt, At)
        torch.cuda.synchronize()
        outs.append((eng.params.clone(), eng.losses_dev.clone()))
    for p, l in outs[1:]:
        assert torch.equal(l, outs[0][1]), (l, outs[0][1])
        assert torch.equal(p, outs[0][0]), (p - outs[0][0]).abs().max()


def test_full_size_step_vs_oracle_and_replay_modes():
    """BASELINE configs[1] at full size (224x224 pair, ViT-B/8, T = 785): the first step (CLS warm-up regime + the
    entire-image branch) against the fp32 CPU oracle at identical parameters -- every loss term within 3e-2, whole-arena
    generator gradient within 5e-2 rel-L2 -- and, as the size-independent property, graph replay == eager single-stream
    launches bit for bit over 3 steps."""
    from oracle import dino_vit
    from oracle.step import SpliceOracle
    from splice_amd import _lib
    from splice_amd.engine import SpliceEngine
    cfg = dict(dino_model_name="dino_vitb8", dino_global_patch_size=224)
    A, B = synth.smooth_image_pair(123, 0, 224, 224)
    vit_state = synth.vit_params(7, "dino_vitb8", img_size=224, w_std=0.03)
    gen_state = synth.generator_params(9, 0.02)
    eng = SpliceEngine(cfg, vit_state, gen_state, (224, 224), (224, 224))
    m = dino_vit.VisionTransformer(8, 768, 12, 12, img_size=224).eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in vit_state.items()})
    orc = SpliceOracle(m, {k: torch.from_numpy(v) for k, v in gen_state.items()}, cfg)
    At, Bt = torch.from_numpy(A), torch.from_numpy(B)
    Ad, Bd = At.to(DEV), Bt.to(DEV)
    lo, _, og = orc.step(At[None], Bt[None], At[None])
    eng.step(Ad, Bd, Ad)
    le = eng.losses()
    assert set(le) == set(lo)
    for k in lo:
        assert abs(le[k] - lo[k]) / abs(lo[k]) < 3e-2, (k, le[k], lo[k])
    num = den = 0.0
    for (name, gt), go in zip(eng.gen.unflatten(eng.grads).items(), og):
        if name.endswith("0.bias") and name != "9.0.bias":   # conv biases feeding a BatchNorm: analytically zero
            continue
        d = (gt.cpu().double() - go.reshape(-1).double()).norm().item()
        num, den = num + d * d, den + go.double().norm().item() ** 2
    assert (num / den) ** 0.5 < 5e-2, (num / den) ** 0.5
    # replay modes at full size
    ref = None
    for graph, overlap in ((1, 1), (0, 0)):
        e2 = SpliceEngine(cfg, None, gen_state, (224, 224), (224, 224), vit_engine=eng.vit)
        _lib.check(_lib.lib().splice_step_use_graph(e2.handle, graph))
        _lib.check(_lib.lib().splice_step_use_overlap(e2.handle, overlap))
        for _ in range(3):
            e2.step(Ad, Bd, Ad)
        torch.cuda.synchronize()
        if ref is None:
            ref = e2.params.clone()
        else:
            assert torch.equal(e2.params, ref)
        del e2


@pytest.mark.parametrize("name,patch,dim,heads", [("dino_vits16", 16, 384, 6), ("dino_vitb16", 16, 768, 12), ("dino_vits8", 8, 384, 6)])
def test_first_step_vs_oracle_other_dino_variants(name, patch, dim, heads):
    """The other three DINO variants the reference's `dino_model_name` accepts (models/extractor.py:20), 224x224 pair: first
    step (CLS warm-up + entire-image branch) against the fp32 CPU oracle, same bars as the ViT-B/8 test above."""
    from oracle import dino_vit
    from oracle.step import SpliceOracle
    from splice_amd.engine import SpliceEngine
    cfg = dict(dino_model_name=name, dino_global_patch_size=224)
    A, B = synth.smooth_image_pair(321, 0, 224, 224)
    vit_state = synth.vit_params(7, name, img_size=224, w_std=0.03)
    gen_state = synth.generator_params(9, 0.02)
    eng = SpliceEngine(cfg, vit_state, gen_state, (224, 224), (224, 224))
    m = dino_vit.VisionTransformer(patch, dim, 12, heads, img_size=224).eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in vit_state.items()})
    orc = SpliceOracle(m, {k: torch.from_numpy(v) for k, v in gen_state.items()}, cfg)
    At, Bt = torch.from_numpy(A), torch.from_numpy(B)
    lo, _, og = orc.step(At[None], Bt[None], At[None])
    eng.step(At.to(DEV), Bt.to(DEV), At.to(DEV))
    le = eng.losses()
    assert set(le) == set(lo)
    for k in lo:
        assert abs(le[k] - lo[k]) / abs(lo[k]) < 3e-2, (k, le[k], lo[k])
    num = den = 0.0
    for (pname, gt), go in zip(eng.gen.unflatten(eng.grads).items(), og):
        if pname.endswith("0.bias") and pname != "9.0.bias":
            continue
        d = (gt.cpu().double() - go.reshape(-1).double()).norm().item()
        num, den = num + d * d, den + go.double().norm().item() ** 2
    assert (num / den) ** 0.5 < 5e-2, (num / den) ** 0.5


def test_large_size_step_replay_modes_and_vit_parity():
    """BASELINE configs[3] (448x448 pair, ViT-B/8, T = 3137): the long-sequence kernel variants (32 queries per wave,
    separate dQ / dK-dV launches, 128-wide GEMM tiles, interpolated position table).  Size-independent properties:
    graph replay == eager single-stream launches bit for bit; losses finite and decreasing over the CLS warm-up; and the
    layer-11 keys of the generated image against the fp32 oracle ViT on the same pixels (2e-2 rel-L2)."""
    from oracle import dino_vit
    from oracle import extractor as oext
    from splice_amd import _lib
    from splice_amd.engine import SpliceEngine
    from splice_amd.vit import KIND_QKV_LAST_F32
    cfg = dict(dino_model_name="dino_vitb8", dino_global_patch_size=448, entire_A_every=10 ** 9)
    A, B = synth.smooth_image_pair(321, 0, 448, 448)
    vit_state = synth.vit_params(7, "dino_vitb8", img_size=224, w_std=0.03)
    gen_state = synth.generator_params(9, 0.02)
    Ad, Bd = torch.from_numpy(A).to(DEV), torch.from_numpy(B).to(DEV)
    ref, vit = None, None
    for graph, overlap in ((1, 1), (0, 0)):
        eng = SpliceEngine(cfg, vit_state if vit is None else None, gen_state, (448, 448), None, vit_engine=vit)
        vit = eng.vit
        _lib.check(_lib.lib().splice_step_use_graph(eng.handle, graph))
        _lib.check(_lib.lib().splice_step_use_overlap(eng.handle, overlap))
        ls = []
        for _ in range(3):
            eng.step(Ad, Bd, None)
            ls.append(eng.losses()["loss"])
        torch.cuda.synchronize()
        assert all(np.isfinite(ls)) and ls[2] < ls[0]
        if ref is None:
            ref = eng.params.clone()
            # keys of pass 2 (x' = G(A)) of the last forward vs the oracle ViT applied to the same generated image
            x = eng.generate(Ad[None])   # parameters AFTER the 3rd update: compare on a fresh forward of both sides
            ctx = vit.context(1, 448, 448, need_grad=False)
            ctx.forward(x, normalize=True)
            qkv = ctx.read(KIND_QKV_LAST_F32, 11)[0, : ctx.T].cpu()
            m = dino_vit.VisionTransformer(8, 768, 12, 12, img_size=224).eval()
            m.load_state_dict({k: torch.from_numpy(v) for k, v in vit_state.items()})
            mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
            std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
            with torch.no_grad():
                k_ref = oext.keys_from_input(m, (x.cpu() - mean) / std, 11)          # [h, T, d]
            k_got = qkv[:, 768:1536].reshape(ctx.T, 12, 64).permute(1, 0, 2)
            rel = ((k_got.double() - k_ref.double()).norm() / k_ref.double().norm()).item()
            assert rel < 2e-2, rel
        else:
            assert torch.equal(eng.params, ref)
        del eng
