"""GPU parity of the fused optimisation step (C ABI splice_step_*) against the loss trajectories
recorded from the REFERENCE loop (Model + LossG + Adam, oracle/make_golden.py) and the fp32 oracle.

Tolerances (bf16 ViT, fp32 generator/losses/Adam; 2-3 x the measured deviations, profiles/r06_step_tests_verbose.txt --
teacher-forced losses 1.4e-3 .. 2.8e-3 measured, bar 1e-2; whole-arena generator gradient 4e-3 .. 1.8e-2 measured, bar 3e-2; the
outlier-weight / real-checkpoint tests keep 2e-2 / 3e-2, the n_crops batches 1e-2 / 6e-2): every entry of the loss dict within 1e-2
relative at steps 0-2 (identical parameters on both sides up to one or two updates).

After that NO implementation can be compared pointwise, the reference included: Adam with beta1 = 0 moves every parameter by ~lr whatever
the size of its gradient component, so rounding-level differences in near-zero components re-route the run.  Measured with the fp32 CPU
oracle (oracle/trajectory_ensemble.py, tests/golden/trajectory_ensemble.json): the SAME fp32 code with one thread against many threads is
above 2 % from step 6, up to 84 % apart, 20.7 dB between the two final images.  So the free run is checked two ways:
  * as a member of a family -- its 6-step window means, the level it reaches and its final image (PSNR, channel statistics) must lie inside
    the range of that ensemble: the fp32 loop with another thread count, with gradient noise of relative size 1e-2 / 2e-2 / 6e-2 (8 seeds each:
    the engine's measured error along the run is 0.7e-2 .. 6.8e-2) and -- since half of that error is no noise but the bf16 rounding of the
    frozen ViT weights -- the fp32 loop run ON the rounded weights; widened by 10 %;
  * pointwise where pointwise MEANS something -- at 15 steps of the free run the fp32 oracle is evaluated at the engine's own parameters of
    that step: every reported loss entry within 1e-2 (measured 4.3e-3), and the gradient the engine descended along within 1.4e-1 (cosine
    above 0.996) of the fp32 oracle's and within 6e-2 of the fp32 oracle's on the engine's rounded weights -- ASSERTED (round 5 only recorded
    them); see the comment above TRAJ_GRAD_BAR_FP32 for why the relative error grows along the run.  A bias in a loss term or in the direction
    of descent fails there whatever the trajectory does.
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


def _oracle_for(name, img_size, vit_state, gen_state, cfg, bf16_weights=False):
    from oracle import dino_vit
    from oracle.step import SpliceOracle
    patch, dim, depth, heads = dino_vit.DINO_CONFIGS[name]
    m = dino_vit.VisionTransformer(patch, dim, depth, heads, img_size=img_size).eval()
    if bf16_weights:   # the fp32 oracle on the bf16-rounded weights the engine holds (oracle/dino_vit.py round_weights_bf16)
        m.load_state_dict(dino_vit.round_weights_bf16(vit_state, dim))
    else:
        m.load_state_dict({k: torch.from_numpy(v) for k, v in vit_state.items()})
    return SpliceOracle(m, {k: torch.from_numpy(v) for k, v in gen_state.items()}, cfg)


def _grad_rel_err(eng, og, with_cos=False):
    """whole-arena relative L2 distance of the engine's generator gradient to the oracle's autograd gradient (with_cos: and their cosine)"""
    num = den = dot = ne = 0.0
    for (name, gt), go in zip(eng.gen.unflatten(eng.grads).items(), og):
        if name.endswith("0.bias") and name != "9.0.bias":
            continue   # conv biases that feed a train-mode BatchNorm: analytically zero (the oracle holds rounding noise)
        a, b = gt.cpu().double(), go.reshape(-1).double()
        d = (a - b).norm().item()
        num, den = num + d * d, den + b.norm().item() ** 2
        dot, ne = dot + float(a @ b), ne + a.norm().item() ** 2
    rel = (num / den) ** 0.5
    return (rel, dot / (ne * den) ** 0.5) if with_cos else rel


def _teacher_forced(eng, orc, A, B, A_ent, steps, loss_tol=1e-2, grad_tol=3e-2, tag=""):
    """Before every step the engine's parameters are reset to the oracle's, so both sides evaluate the SAME point
    (see test_step_gradients_vs_oracle_teacher_forced for why); every loss entry and the whole gradient are compared."""
    At, Bt = torch.from_numpy(A), torch.from_numpy(B)
    Et = None if A_ent is None else torch.from_numpy(A_ent)
    Ad, Bd, Ed = At.to(DEV), Bt.to(DEV), None if Et is None else Et.to(DEV)
    worst_l = worst_g = 0.0
    for step in range(steps):
        eng.params.copy_(eng.gen.flatten({k: v.detach() for k, v in orc.params.items()}))
        lo, _, og = orc.step(At[None], Bt[None], None if Et is None else Et[None])
        eng.step(Ad, Bd, Ed)
        le = eng.losses()
        assert set(le) == set(lo), (step, sorted(le), sorted(lo))
        for k in lo:
            rel = abs(le[k] - lo[k]) / abs(lo[k])
            worst_l = max(worst_l, rel)
            assert rel < loss_tol, (tag, step, k, le[k], lo[k])
        rel = _grad_rel_err(eng, og)
        worst_g = max(worst_g, rel)
        print(f"    {tag} step {step}: loss {le['loss']:.4f} vs oracle {lo['loss']:.4f}; gradient rel err {rel:.3e}")
        assert rel < grad_tol, (tag, step, rel)
    return worst_l, worst_g


# Bars of the spot checks along the free run (round 6; measured: profiles/r06_traj_grad_error.txt, r06_step_tests_verbose.txt).  The engine's
# gradient error against the fp32 oracle is a FLOOR, not a fraction: |g_e - g_o| stays at 7 .. 15 (absolute, this fixture) while |g_o| falls
# from 1470 to 135 along the run (log-log slope 0.24), so the RELATIVE error grows from 0.7e-2 to 6.8e-2 where the gradient is smallest (step 36).
# The floor has two parts of similar size: the bf16 rounding of the frozen ViT WEIGHTS -- a fixed perturbation of the model, the same at every
# step (which is why the error vectors of neighbouring steps are correlated, cosine up to 0.87: it is not isotropic noise) -- 0.3e-2 .. 5.7e-2,
# and the rounding of activations / probabilities, 0.4e-2 .. 2.5e-2, measured against the fp32 oracle evaluated ON the rounded weights.
TRAJ_GRAD_BAR_FP32 = 1.4e-1       # 2 x 6.8e-2: against the fp32 oracle (fp32 weights)
TRAJ_GRAD_BAR_BF16W = 6e-2        # 2 x 2.8e-2 (step 75, an entire-image step; 2.5e-2 at step 36): against the fp32 oracle on the engine's bf16-rounded weights
TRAJ_GRAD_COS = 0.996             # measured >= 0.99809


def _run_with_spot_checks(eng, A, B, n, spots, vit_name="dino_vits8", img_size=64, vit_seed=7, rtol=1e-2):
    """``_run`` + at every step in ``spots`` the fp32 oracle evaluated AT THE ENGINE'S PARAMETERS of that step (free-running engine,
    no re-synchronisation of the engine): the reported loss entries must be the true ones, and so must the GRADIENT the engine took its step
    along -- the engine's gradient arena after the step against the oracle's autograd gradient at the same point, whole-arena relative L2 and
    cosine, against BOTH oracles (fp32 weights: bar 1.4e-1 / cos 0.996; the bf16-rounded weights the engine holds: bar 6e-2).  A bias anywhere
    along the run -- in a loss term or in the direction of descent -- fails here whatever the chaotic trajectory does."""
    from oracle import losses as OL
    vit_state = synth.vit_params(vit_seed, vit_name, img_size=img_size, w_std=0.05)
    orc = _oracle_for(vit_name, img_size, vit_state, synth.generator_params(1, 0.02), eng.cfg)
    orc_w = _oracle_for(vit_name, img_size, vit_state, synth.generator_params(1, 0.02), eng.cfg, bf16_weights=True)
    rows = []
    At, Bt = torch.from_numpy(A), torch.from_numpy(B)
    Ad, Bd = At.to(DEV), Bt.to(DEV)
    worst = worst_g = worst_w = 0.0
    worst_cos = 1.0

    def at_snapshot(o, snap, step):
        with torch.no_grad():
            for k, v in o.params.items():
                v.copy_(snap[k].cpu().reshape(v.shape))
        o.step_idx = step - 1
        o.lambdas = OL.initial_lambdas(o.cfg)
        if step >= o.cfg["cls_warmup"]:
            OL.update_lambdas(o.lambdas, o.cfg, o.cfg["cls_warmup"])
        return o.step(At[None], Bt[None], At[None])
    for step in range(n):
        if step in spots:
            snap = eng.gen.unflatten(eng.params.clone())
        eng.step(Ad, Bd, Ad)
        rows.append(eng.losses())
        if step in spots:
            lo, _, og = at_snapshot(orc, snap, step)
            le = rows[-1]
            assert set(le) == set(lo), (step, sorted(le), sorted(lo))
            for k in lo:
                rel = abs(le[k] - lo[k]) / abs(lo[k])
                worst = max(worst, rel)
                assert rel < rtol, ("reported loss != oracle loss at the engine's own parameters", step, k, le[k], lo[k])
            grel, gcos = _grad_rel_err(eng, og, with_cos=True)
            _, _, ogw = at_snapshot(orc_w, snap, step)
            grel_w = _grad_rel_err(eng, ogw)
            worst_g, worst_w, worst_cos = max(worst_g, grel), max(worst_w, grel_w), min(worst_cos, gcos)
            print(f"    free-running step {step}: reported loss {le['loss']:.3f}, fp32 oracle at the same parameters {lo['loss']:.3f}; gradient rel err {grel:.3e} "
                  f"(cos {gcos:.5f}); against the oracle on the bf16-rounded weights {grel_w:.3e}")
            assert grel < TRAJ_GRAD_BAR_FP32 and gcos > TRAJ_GRAD_COS, ("gradient != fp32 oracle gradient at the engine's own parameters", step, grel, gcos)
            assert grel_w < TRAJ_GRAD_BAR_BF16W, ("gradient != gradient of the fp32 oracle on the engine's bf16-rounded weights", step, grel_w)
    print(f"    along the free-running trajectory (steps {sorted(spots)}): reported-vs-true loss worst rel {worst:.3e}, gradient worst rel {worst_g:.3e} "
          f"(cos >= {worst_cos:.5f}), against the bf16-weight oracle {worst_w:.3e}")
    return rows


def test_trajectory_a_identity_resize(golden_dir):
    g = np.load(os.path.join(golden_dir, "steps.npz"))
    keys = [str(k) for k in g["loss_keys"]]
    assert keys == LOSS_KEYS
    A, B = synth.smooth_image_pair(32, 0, 64, 64)
    eng = _engine({}, A, B, 31, 64)
    rows = _run_with_spot_checks(eng, A, B, 78, spots={1, 6, 12, 18, 24, 30, 36, 42, 48, 54, 60, 66, 72, 75, 77})
    _check(rows, g["a/losses"], keys, 0, 3, 1e-2)
    mine = np.array([r["loss"] for r in rows])
    ref = g["a/losses"][:, 0]
    assert np.isfinite(mine).all()
    # Beyond the first steps the run is compared as a MEMBER OF A FAMILY, not pointwise: tests/golden/trajectory_ensemble.json (oracle/trajectory_ensemble.py)
    # holds the reference loop re-run in fp32 on the CPU with another thread count and with gradient noise of the engine's measured size
    # (eps = 1e-2, 2e-2; 8 seeds each).  The reference is not reproducible against ITSELF pointwise (1 thread vs many: > 2 % from step 8,
    # 20 dB between the final images), so the bars below are the family's own range, widened by 10 %.
    import json
    fam = json.load(open(os.path.join(golden_dir, "trajectory_ensemble.json")))
    env = fam["envelope"]
    print(f"    the fp32 reference against itself (1 thread vs many): first step above 2 % = {fam['fp32_self_reproducibility']['first_step_above_2_percent']}, "
          f"PSNR between the final images {fam['fp32_self_reproducibility']['psnr_db_between_final_images']:.1f} dB")
    for lo in range(1, 73, 6):     # 6-step window means (step 0 excluded: 3x larger, checked exactly above; step 75 = an entire-image step)
        a, b = mine[lo:lo + 6].mean(), ref[lo:lo + 6].mean()
        lo_b, hi_b = env["window_ratio"][str(lo)]
        print(f"    steps {lo}..{lo + 5}: mean loss {a:.1f} vs reference {b:.1f} (ratio {a / b:.2f}; family {lo_b:.2f} .. {hi_b:.2f})")
        assert 0.9 * lo_b <= a / b <= 1.1 * hi_b, (lo, a / b, lo_b, hi_b)
    tail_mine, tail_ref = np.sort(mine[60:75])[:5].mean(), np.sort(ref[60:75])[:5].mean()
    print(f"    level reached (steps 60..74): {tail_mine:.1f} vs reference {tail_ref:.1f} (ratio {tail_mine / tail_ref:.2f}; family {env['level_ratio'][0]:.2f} .. {env['level_ratio'][1]:.2f})")
    assert 0.9 * env["level_ratio"][0] <= tail_mine / tail_ref <= 1.1 * env["level_ratio"][1]
    out = eng.generate(torch.from_numpy(A)[None].to(DEV)).cpu().numpy()
    refimg = g["a/final_out"]
    mse = float(((out - refimg) ** 2).mean())
    psnr = 10 * np.log10(1.0 / max(mse, 1e-12))
    print(f"    final image PSNR(HIP engine vs reference CPU fp32) after 78 steps: {psnr:.1f} dB (family {env['psnr_db'][0]:.1f} .. {env['psnr_db'][1]:.1f} dB)")
    assert out.shape == refimg.shape and np.isfinite(out).all() and 0.0 <= out.min() and out.max() <= 1.0
    # output PIXELS (north_star: "loss trajectories and output pixels"): no member of the family reproduces the reference's image pointwise
    # (15 .. 21 dB); the image must be as close as the family's members are, and its per-channel statistics inside their range
    assert psnr >= env["psnr_db"][0] - 1.0, (psnr, env["psnr_db"])
    for c in range(3):
        mo, so = float(out[0, c].mean()), float(out[0, c].std())
        (mlo, mhi), (slo, shi) = env["channel_mean"][c], env["channel_std"][c]
        print(f"    channel {c}: mean {mo:.4f} (family {mlo:.4f} .. {mhi:.4f}), std {so:.4f} (family {slo:.4f} .. {shi:.4f})")
        assert mlo - 0.02 <= mo <= mhi + 0.02 and 0.8 * slo <= so <= 1.25 * shi, (c, mo, so)
    lum_a = A.mean(0)
    lum_o = out[0].mean(0)
    corr = float(np.corrcoef(lum_a.reshape(-1), lum_o.reshape(-1))[0, 1])
    print(f"    luminance correlation with the structure image: {corr:.3f}")


def test_trajectory_b_resize_nonsquare(golden_dir):
    """48x80 pair: Resize 48->64 (non-identity, differentiable) and an 8x13 token grid (interpolated
    position embedding)."""
    g = np.load(os.path.join(golden_dir, "steps.npz"))
    keys = [str(k) for k in g["loss_keys"]]
    A, B = synth.smooth_image_pair(34, 1, 48, 80)
    eng = _engine({}, A, B, 33, 64)
    assert eng.vit_hw == (64, 106)
    rows = _run(eng, A, B, 4)
    _check(rows, g["b/losses"], keys, 0, 2, 1e-2)     # measured 1.9e-3
    _check(rows, g["b/losses"], keys, 2, 4, 3e-2)     # measured 9.9e-3 (two and three free-running updates behind the fixture; round 5 carried 2.5e-1 here)


def test_step_gradients_vs_oracle_teacher_forced():
    """Steps 0..4 (all lambda regimes: step 0 = entire + cls, steps >= 1 = ssim + cls + id) with the
    engine's parameters re-synchronised to the oracle's before every step, so both sides evaluate
    the SAME point: every loss entry within 1e-2 and the whole generator gradient within 3e-2
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
            assert abs(le[k] - lo[k]) / abs(lo[k]) < 1e-2, (step, k, le[k], lo[k])
        got = eng.gen.unflatten(eng.grads)
        num = den = 0.0
        for (name, gt), go in zip(got.items(), og):
            if name.endswith("0.bias") and name != "9.0.bias":
                continue  # zero-gradient conv biases (feed a BatchNorm)
            d = (gt.cpu().double() - go.reshape(-1).double()).norm().item()
            num, den = num + d * d, den + go.double().norm().item() ** 2
        rel = (num / den) ** 0.5
        print(f"    step {step}: loss {le['loss']:.3f} vs {lo['loss']:.3f}; generator-gradient rel err vs oracle {rel:.3e}")
        assert rel < 3e-2, rel


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


def test_graph_executables_are_updated_in_place_across_crop_sizes_and_handles():
    """A captured step executable is never destroyed in a running process (round 5) and -- round 6 -- the pool it retires to is keyed by the
    SIGNATURE of the captured graph (device, kernel function of every node in order, edge list): the next capture with the same launch sequence,
    by this engine or by the next pair's engine, updates it in place (hipGraphExecUpdate).  Runs of equal crop sizes (each long enough to
    capture) at several sizes, then a second engine; the parameters must equal the eager run's BIT FOR BIT and the runtime may refuse NO update
    (round 5 keyed the pool by a few configuration words and saw refusals outnumber updates here, VERDICT r5 "What's weak" #5)."""
    import ctypes as C
    from splice_amd import _lib
    A, B = synth.smooth_image_pair(78, 0, 64, 64)
    At, Bt = torch.from_numpy(A).to(DEV), torch.from_numpy(B).to(DEV)
    sizes = [64] * 5 + [63] * 5 + [62] * 4 + [64] * 4 + [61] * 1 + [63] * 4

    def run(graph):
        eng = _engine(dict(cls_warmup=1, entire_A_every=1000), A, B, gen_seed=6, img_size=64)
        _lib.check(_lib.lib().splice_step_use_graph(eng.handle, graph))
        for sz in sizes:
            eng.step(At[:, :sz, :sz].contiguous(), Bt[:, :sz, :sz].contiguous(), At)
        torch.cuda.synchronize()
        st = (C.c_longlong * 3)()
        _lib.check(_lib.lib().splice_step_graph_stats(eng.handle, st))
        return eng.params.clone(), list(st)
    p_eager, st0 = run(0)
    assert st0 == [0, 0, 0], st0
    p_graph, st1 = run(1)
    assert torch.equal(p_graph, p_eager)
    p_graph2, st2 = run(1)     # a second handle: its captures find the first one's executables in the pool
    assert torch.equal(p_graph2, p_eager)
    print(f"    graph executables: first engine updates / refusals / instantiations {st1}, second engine {st2}")
    assert st1[1] == 0 and st2[1] == 0, (st1, st2)              # same signature => the runtime accepts the update
    assert st1[0] + st1[2] == 5, st1                            # the ordinary regime captured at 64, 63, 62, 64, 63
    assert st2[0] + st2[2] == 5 and st2[2] <= st1[2], (st1, st2)   # the second engine instantiates nothing the first one has not left in the pool
    assert st2[0] >= 1, st2


def test_step_phases_and_follower_are_exact():
    """splice_step_set_phases: (a) a step run as two calls (generator forward + ViT part, then generator backward + Adam)
    launches the same kernels as the one-call step: parameters and losses agree BIT FOR BIT over 5 steps (entire-image
    branch, warm-up switch); (b) a follower at the leader's own scale adds the same image gradient again, and every
    backward kernel of the generator is linear in it: the summed parameter gradient is exactly twice the single one."""
    from splice_amd import _lib
    L = _lib.lib()
    A, B = synth.smooth_image_pair(78, 0, 64, 64)
    At, Bt = (torch.from_numpy(x).to(DEV) for x in (A, B))
    cfg = dict(cls_warmup=2, entire_A_every=4)
    outs = []
    for split in (False, True):
        eng = _engine(cfg, A, B, gen_seed=6, img_size=64)
        for i in range(5):
            if split:
                _lib.check(L.splice_step_set_phases(eng.handle, 3, None))
                eng.step(At, Bt, At)
                _lib.check(L.splice_step_set_phases(eng.handle, 4, None))
                eng.step(At, Bt, At, _repeat=True)
            else:
                eng.step(At, Bt, At)
        torch.cuda.synchronize()
        outs.append((eng.params.clone(), eng.losses_dev.clone()))
    assert torch.equal(outs[0][1], outs[1][1])
    assert torch.equal(outs[0][0], outs[1][0]), (outs[0][0] - outs[1][0]).abs().max()
    # (b) leader + follower at one scale: gradient-only mode, warm-up over (every term on), entire-image step
    cfg = dict(cls_warmup=0, entire_A_every=4)
    single = _engine(cfg, A, B, gen_seed=6, img_size=64)
    _lib.check(L.splice_step_set_mode(single.handle, 1, 0))
    single.step(At, Bt, At)
    lead, foll = _engine(cfg, A, B, gen_seed=6, img_size=64), _engine(cfg, A, B, gen_seed=6, img_size=64)
    _lib.check(L.splice_step_set_mode(lead.handle, 1, 0))
    _lib.check(L.splice_step_set_mode(foll.handle, 1, 0))
    _lib.check(L.splice_step_set_phases(foll.handle, 2, lead.handle))
    assert L.splice_step_set_phases(foll.handle, 3, lead.handle) != 0      # a follower runs the ViT part only
    assert L.splice_step_set_phases(lead.handle, 2, foll.handle) != 0      # no chains of followers
    _lib.check(L.splice_step_set_phases(lead.handle, 3, None))
    lead.step(At, Bt, At)
    foll.step(At, Bt, At)
    _lib.check(L.splice_step_set_phases(lead.handle, 4, None))
    lead.step(At, Bt, At, _repeat=True)
    torch.cuda.synchronize()
    assert torch.equal(foll.losses_dev, single.losses_dev) and torch.equal(lead.losses_dev, single.losses_dev)
    assert float(single.grads.abs().sum()) > 0
    assert torch.equal(lead.grads, 2 * single.grads), (lead.grads - 2 * single.grads).abs().max()


def test_full_size_step_vs_oracle_and_replay_modes():
    """BASELINE configs[1] at full size (224x224 pair, ViT-B/8, T = 785), teacher-forced steps 0, 1, 2 against the fp32 CPU
    oracle: step 0 is the CLS warm-up regime + the entire-image branch, steps 1-2 are the ORDINARY regime every timed step
    of bench.py runs (global ssim + cls + id, two N=1 generator plans, split-K dgrads, id-loss seeds at T = 785).  Every
    loss term within 1e-2, whole-arena generator gradient within 3e-2 rel-L2 (bf16 ViT; synthetic N(0, 0.03) weights -- a
    trained checkpoint has outlier channels, see test_outlier_weights_step_vs_oracle).  Then the size-independent property:
    graph replay == eager single-stream launches bit for bit over 4 steps."""
    from splice_amd import _lib
    from splice_amd.engine import SpliceEngine
    cfg = dict(dino_model_name="dino_vitb8", dino_global_patch_size=224)
    A, B = synth.smooth_image_pair(123, 0, 224, 224)
    vit_state = synth.vit_params(7, "dino_vitb8", img_size=224, w_std=0.03)
    gen_state = synth.generator_params(9, 0.02)
    eng = SpliceEngine(cfg, vit_state, gen_state, (224, 224), (224, 224))
    orc = _oracle_for("dino_vitb8", 224, vit_state, gen_state, cfg)
    _teacher_forced(eng, orc, A, B, A, 3, tag="224/B8")
    Ad, Bd = torch.from_numpy(A).to(DEV), torch.from_numpy(B).to(DEV)
    # replay modes at full size
    ref = None
    for graph, overlap in ((1, 1), (0, 0)):
        e2 = SpliceEngine(cfg, None, gen_state, (224, 224), (224, 224), vit_engine=eng.vit)
        _lib.check(_lib.lib().splice_step_use_graph(e2.handle, graph))
        _lib.check(_lib.lib().splice_step_use_overlap(e2.handle, overlap))
        for _ in range(4):   # (a graph is captured at the third step with identical shapes and replayed at the fourth)
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
        assert abs(le[k] - lo[k]) / abs(lo[k]) < 1e-2, (k, le[k], lo[k])
    num = den = 0.0
    for (pname, gt), go in zip(eng.gen.unflatten(eng.grads).items(), og):
        if pname.endswith("0.bias") and pname != "9.0.bias":
            continue
        d = (gt.cpu().double() - go.reshape(-1).double()).norm().item()
        num, den = num + d * d, den + go.double().norm().item() ** 2
    assert (num / den) ** 0.5 < 3e-2, (num / den) ** 0.5


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
        for _ in range(4):   # (graph capture at the third step, replay at the fourth)
            eng.step(Ad, Bd, None)
            ls.append(eng.losses()["loss"])
        torch.cuda.synchronize()
        assert all(np.isfinite(ls)) and ls[2] < ls[0]
        if ref is None:
            ref = eng.params.clone()
            # keys of pass 2 (x' = G(A)) of the last forward vs the oracle ViT applied to the same generated image
            x = eng.generate(Ad[None])   # parameters AFTER the last update: compare on a fresh forward of both sides
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


def test_config0_128px_vitb16_resize_vs_oracle():
    """BASELINE configs[0]: 128x128 pair, DINO ViT-B/16, dino_global_patch_size 224 -- every ViT input goes through the
    NON-identity bilinear Resize 128 -> 224 (and its adjoint in the backward), T = 197.  Teacher-forced steps 0-1 (entire +
    cls, then ssim + cls + id) against the fp32 oracle: losses 1e-2, gradient 3e-2."""
    cfg = dict(dino_model_name="dino_vitb16", dino_global_patch_size=224)
    A, B = synth.smooth_image_pair(128, 0, 128, 128)
    vit_state = synth.vit_params(7, "dino_vitb16", img_size=224, w_std=0.03)
    gen_state = synth.generator_params(9, 0.02)
    eng = SpliceEngine(cfg, vit_state, gen_state, (128, 128), (128, 128))
    assert eng.vit_hw == (224, 224) and eng.ctx_g.T == 197
    orc = _oracle_for("dino_vitb16", 224, vit_state, gen_state, cfg)
    _teacher_forced(eng, orc, A, B, A, 2, tag="128->224/B16")


def test_entire_image_hits_max_size_cap_and_ragged_patch_grid_vs_oracle():
    """Edge of util/losses.py:19-24: ``Resize(224, max_size=480)`` on a WIDE entire image -- 64x150 -> 224x525 exceeds the cap
    and becomes 204x480, which is not a multiple of the patch size either (DINO's patch-embed convolution floors: 25 x 60
    patches, position embedding interpolated to that non-square grid).  Crops are 64x64 up-sampled to 224.  Steps 0 (entire
    self-similarity + entire / global CLS on the capped input) and 1 (ordinary) against the fp32 oracle, teacher-forced."""
    from splice_amd.engine import resize_output_size
    assert resize_output_size(64, 150, 224, 480) == (204, 480)
    name = "dino_vits8"
    cfg = dict(dino_model_name=name, dino_global_patch_size=224)
    vit_state = synth.vit_params(7, name, img_size=224, w_std=0.05)
    gen_state = synth.generator_params(9, 0.02)
    A = synth.uniform(41, "cap/A", (3, 64, 64))
    B = synth.uniform(41, "cap/B", (3, 64, 64))
    E = synth.uniform(41, "cap/E", (3, 64, 150))
    eng = SpliceEngine(cfg, vit_state, gen_state, (64, 64), (64, 150))
    assert eng.ctx_e.T == 25 * 60 + 1
    orc = _oracle_for(name, 224, vit_state, gen_state, cfg)
    wl, wg = _teacher_forced(eng, orc, A, B, E, 2, tag="max_size cap")
    print(f"    max_size-cap edge: worst loss rel {wl:.3e}, worst gradient rel {wg:.3e}")


def test_reference_default_shape_900x1200_crops_855_to_900_vs_oracle():
    """The reference's own default workload shape (VERDICT r4 missing #2): ``conf/default/config.yaml:3,5-6`` + ``data/Dataset.py:44-51`` load
    1200 x 900 images with ``A_resize: -1``; ``data/transforms.py:21-23`` draws square global crops of 0.95 h .. h = 855 .. 900 pixels,
    independently for A and B; ``util/losses.py:19-24`` resizes every crop to 224 x 224 (a 3.8 - 4 x bilinear down-scale WITHOUT antialias)
    and the entire 900 x 1200 image to 224 x 298 (37 x 28 patches, T = 1037, non-square position table).  Generator planes of 0.73 - 1.08 MP,
    ViT-B/8.  Teacher-forced against the fp32 CPU oracle: step 0 (CLS warm-up + the entire-image branch, crops 872 / 861) and step 1
    (ordinary regime, crops 900 / 855 -- the extremes, and a crop-size change between steps), usual bars."""
    from splice_amd.engine import resize_output_size
    assert resize_output_size(900, 1200, 224, 480) == (224, 298)
    name = "dino_vitb8"
    cfg = dict(dino_model_name=name, dino_global_patch_size=224)
    vit_state = synth.vit_params(7, name, img_size=224, w_std=0.03)
    gen_state = synth.generator_params(9, 0.02)
    A, B = synth.smooth_image_pair(900, 0, 900, 1200)
    # + pixel-scale detail: a 4 x down-scale without antialias must alias it exactly as the oracle's (torchvision 0.10 tensor Resize)
    A = (0.75 * A + 0.25 * synth.uniform(900, "default-shape/A", (3, 900, 1200))).astype(np.float32)
    B = (0.75 * B + 0.25 * synth.uniform(900, "default-shape/B", (3, 900, 1200))).astype(np.float32)
    eng = SpliceEngine(cfg, vit_state, gen_state, (900, 900), (900, 1200))
    assert eng.ctx_e.T == 28 * 37 + 1 and eng.ctx_g.T == 785
    orc = _oracle_for(name, 224, vit_state, gen_state, cfg)
    At, Bt = torch.from_numpy(A), torch.from_numpy(B)
    Ed = At.to(DEV).contiguous()
    crops = [((872, 11, 200), (861, 30, 77)), ((900, 0, 150), (855, 45, 300))]   # (size, top, left) of the A and the B crop per step
    worst_l = worst_g = 0.0
    for step, ((sa, ya, xa), (sb_, yb, xb)) in enumerate(crops):
        Ac, Bc = At[:, ya:ya + sa, xa:xa + sa].contiguous(), Bt[:, yb:yb + sb_, xb:xb + sb_].contiguous()
        eng.params.copy_(eng.gen.flatten({k: v.detach() for k, v in orc.params.items()}))
        lo, _, og = orc.step(Ac[None], Bc[None], At[None])
        eng.step(Ac.to(DEV), Bc.to(DEV), Ed)
        le = eng.losses()
        assert set(le) == set(lo), (step, sorted(le), sorted(lo))
        for k in lo:
            rel = abs(le[k] - lo[k]) / abs(lo[k])
            worst_l = max(worst_l, rel)
            assert rel < 1e-2, (step, k, le[k], lo[k])
        rel = _grad_rel_err(eng, og)
        worst_g = max(worst_g, rel)
        print(f"    default shape, step {step}: crops {sa} / {sb_}, loss {le['loss']:.4f} vs oracle {lo['loss']:.4f}; gradient rel err {rel:.3e}")
        assert rel < 3e-2, (step, rel)
    print(f"    reference default shape: worst loss rel {worst_l:.3e}, worst gradient rel {worst_g:.3e}")


@pytest.mark.parametrize("term", ["cls", "ssim", "id"])
def test_448_step_loss_and_gradient_vs_oracle(term):
    """BASELINE configs[3] (448x448 pair, ViT-B/8, T = 3137: attn_fwd_kernel<2>, the two-launch attention backward, the
    T = 3137 self-similarity GEMMs, interpolated position table), LOSS and whole-arena GRADIENT against the fp32 CPU oracle
    at identical parameters.  One loss term per case (the other lambdas are zero on both sides) so that the oracle's
    autograd holds ONE differentiated ViT pass of 12 x [12, 3137, 3137] probabilities in host memory, not three."""
    lam = dict(lambda_global_cls=0.0, lambda_global_ssim=0.0, lambda_global_identity=0.0, lambda_entire_cls=0.0, lambda_entire_ssim=0.0)
    lam[{"cls": "lambda_global_cls", "ssim": "lambda_global_ssim", "id": "lambda_global_identity"}[term]] = {"cls": 10.0, "ssim": 1.0, "id": 1.0}[term]
    cfg = dict(dino_model_name="dino_vitb8", dino_global_patch_size=448, entire_A_every=10 ** 9, cls_warmup=0, **lam)
    A, B = synth.smooth_image_pair(448, 0, 448, 448)
    vit_state = synth.vit_params(7, "dino_vitb8", img_size=224, w_std=0.03)
    gen_state = synth.generator_params(9, 0.02)
    eng = SpliceEngine(cfg, vit_state, gen_state, (448, 448), None)
    assert eng.ctx_g.T == 3137
    orc = _oracle_for("dino_vitb8", 224, vit_state, gen_state, cfg)
    _teacher_forced(eng, orc, A, B, None, 1, tag=f"448/{term}")


@pytest.mark.parametrize("name,size", [("dino_vits8", 64), ("dino_vitb8", 224)])
def test_outlier_weights_step_vs_oracle(name, size):
    """every other bf16 tolerance in this suite is measured on zero-mean N(0, sigma) synthetic ViTs, while trained
    DINO checkpoints carry massive residual-stream channels, heavy-tailed LayerNorm gains and large key biases (no real
    checkpoint is available offline).  `synth.vit_params_outlier` builds such a set; teacher-forced steps 0-2 against the
    fp32 oracle bound what the bf16 path (bf16 LN outputs / qkv / attention / self-similarity Gram) does to the loss terms
    and to the generator gradient there.  Bars: losses 2e-2, gradient 3e-2 (measured r2: 7e-3 / 1.3e-2 -- no worse than the
    zero-mean synthetic set)."""
    cfg = dict(dino_model_name=name, dino_global_patch_size=size)
    A, B = synth.smooth_image_pair(77, 3, size, size)
    vit_state = synth.vit_params_outlier(7, name, img_size=size)
    gen_state = synth.generator_params(9, 0.02)
    eng = SpliceEngine(cfg, vit_state, gen_state, (size, size), (size, size))
    orc = _oracle_for(name, size, vit_state, gen_state, cfg)
    wl, wg = _teacher_forced(eng, orc, A, B, A, 3, loss_tol=2e-2, grad_tol=3e-2, tag=f"outlier/{name}")
    print(f"    outlier-weight {name}: worst loss rel err {wl:.3e}, worst gradient rel err {wg:.3e}")


@pytest.mark.skipif(not os.environ.get("SPLICE_DINO_CHECKPOINT"), reason="SPLICE_DINO_CHECKPOINT unset: no trained DINO .pth on this box (none can be fetched offline)")
def test_real_dino_checkpoint_steps_vs_oracle():
    """every ViT in the other tests is synthetic (incl. the outlier-statistics stand-in).  With a real DINO
    checkpoint at hand -- ``SPLICE_DINO_CHECKPOINT=/path/dino_vitbase8_pretrain.pth`` (any of the four variants; the variant is
    read from ``SPLICE_DINO_MODEL``, default dino_vitb8) -- the teacher-forced steps 0, 1, 2 of a 224x224 pair run against the
    fp32 CPU oracle loaded from the SAME file, at the bars of the outlier-weights test (losses 2e-2, whole-arena gradient 3e-2)."""
    from splice_amd.checkpoint import load_dino_checkpoint
    from splice_amd.engine import SpliceEngine
    name = os.environ.get("SPLICE_DINO_MODEL", "dino_vitb8")
    got_name, vit_state = load_dino_checkpoint(os.environ["SPLICE_DINO_CHECKPOINT"], name)
    assert got_name == name
    vit_state = {k: (v.numpy() if hasattr(v, "numpy") else v) for k, v in vit_state.items()}
    cfg = dict(dino_model_name=name, dino_global_patch_size=224)
    A, B = synth.smooth_image_pair(123, 0, 224, 224)
    gen_state = synth.generator_params(9, 0.02)
    eng = SpliceEngine(cfg, vit_state, gen_state, (224, 224), (224, 224))
    orc = _oracle_for(name, 224, vit_state, gen_state, cfg)
    _teacher_forced(eng, orc, A, B, A, 3, loss_tol=2e-2, grad_tol=3e-2, tag="real checkpoint")


def test_full_length_run_configs1_2000_steps():
    """BASELINE configs[1] at its own length: ONE 224x224 pair, ViT-B/8 (T = 785), 2000 optimisation steps (`train.py:51-80` x 2000;
    the paper's length, SURVEY 8d cfg 2) on the fused engine -- graph replay on ordinary steps, the entire-image branch every 75th.
    Checks what a long run can break and a 78-step fixture cannot: every sampled loss finite, the optimisation keeps descending
    (window means of the ordinary-step loss non-increasing within 5 % down to the plateau, within 2 x its floor there), the generated image stays inside [0, 1], device memory
    is flat after the first 200 steps, and the step counter / Adam count agree with the number of calls."""
    from splice_amd.engine import synthetic_engine
    A, B = synth.smooth_image_pair(1234, 0, 224, 224)
    eng, _, _ = synthetic_engine(dict(dino_model_name="dino_vitb8", dino_global_patch_size=224), pair_id=0, hw=(224, 224), seed=1234, device=DEV)
    Ad, Bd = torch.from_numpy(A).to(DEV), torch.from_numpy(B).to(DEV)
    samples, mem = {}, {}
    for step in range(2000):
        eng.step(Ad, Bd, Ad)
        if step % 25 == 24 and step % 75 != 0:
            samples[step] = eng.losses()["loss"]                  # (host sync only here)
        if step in (199, 1999):
            torch.cuda.synchronize()
            free, total = torch.cuda.mem_get_info()
            mem[step] = (total - free, torch.cuda.memory_allocated())
    vals = np.array([samples[k] for k in sorted(samples)])
    assert np.isfinite(vals).all() and len(vals) >= 50
    wins = [vals[i:i + 10].mean() for i in range(0, len(vals) - 9, 10)]
    print("    2000 steps at 224^2 / ViT-B/8: window means of the ordinary-step loss: " + ", ".join(f"{w:.2f}" for w in wins))
    # descending: every window below 1.05 x the previous one until the run has reached its plateau (<= 10 % of the first window:
    # ~0.1 after 600 steps, where the loss of this chaotic optimisation fluctuates by tens of per cent between windows -- seen:
    # 0.09, 0.09, 0.07, 0.10 with one build, 0.08, 0.07, 0.07, 0.06 with another); on the plateau it must stay within 2 x its floor
    for a, b in zip(wins, wins[1:]):
        assert b < 1.05 * a or b <= 0.1 * wins[0], wins
    assert all(w <= 2.0 * min(wins[:i + 1]) for i, w in enumerate(wins) if w <= 0.1 * wins[0]), wins
    assert wins[-1] < 0.2 * wins[0], wins                         # and it actually went somewhere
    out = eng.generate(Ad[None])
    assert out.shape == (1, 3, 224, 224) and torch.isfinite(out).all() and 0.0 <= out.min().item() and out.max().item() <= 1.0
    assert out.std().item() > 1e-3                                # not a constant image
    assert eng.step_idx == 1999
    used0, used1 = mem[199][0], mem[1999][0]
    print(f"    device memory in use after 200 / 2000 steps: {used0 / 2**20:.0f} / {used1 / 2**20:.0f} MiB; torch allocator {mem[199][1] / 2**20:.0f} / {mem[1999][1] / 2**20:.0f} MiB")
    assert abs(used1 - used0) < 64 * 2 ** 20 and mem[1999][1] == mem[199][1]


