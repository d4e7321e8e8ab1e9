"""GPU: P image pairs side by side on one engine.  The reference optimises one pair per process
(train.py:34-49); pairs share only the frozen ViT, so P of them ride the same kernel launches here.  The contract is that
batching is INVISIBLE to a pair: its losses, gradients, parameters and BatchNorm buffers are bit-identical to its own
P = 1 run, whichever batch it rides in (no launch policy may depend on P)."""
import ctypes as C

import numpy as np
import pytest
import torch

from splice_amd import _lib, synth
from splice_amd.engine import MultiPairEngine, SpliceEngine
from splice_amd.generator import GeneratorEngine, GeneratorPlan

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("hw", [(64, 64), (96, 130), (224, 224)])
def test_generator_independent_images_match_single(hw):
    """One plan of 3 INDEPENDENT generators (own parameter arenas) == three N = 1 plans: outputs, every parameter gradient
    and the BatchNorm running statistics bit for bit (incl. the split-K / 8-wave launch policies, taken per image)."""
    gen = GeneratorEngine(device=DEV)
    n, P = gen.numel, 3
    stride = (n + 63) // 64 * 64
    H, W = hw
    params = torch.zeros(P * stride, device=DEV)
    x = torch.from_numpy(np.stack([synth.uniform(5, f"mp/x{p}", (3, H, W)) for p in range(P)])).to(DEV)
    dy = torch.from_numpy(np.stack([synth.normal(6, f"mp/dy{p}", (3, H, W)) for p in range(P)])).to(DEV)
    for p in range(P):
        params[p * stride: p * stride + n] = gen.flatten(synth.generator_params(50 + p, 0.02, perturb_bias=0.05))
    multi = GeneratorPlan(gen, P, H, W, True, stride)
    y = multi.forward(params, x)
    g = multi.backward(params, dy)
    run = torch.zeros(P, gen.buffer_numel, device=DEV)
    plans = (C.c_void_p * 1)(multi.handle)
    _lib.check(_lib.lib().splice_gen_running_stats_update(plans, 1, _lib.ptr(run), run.stride(0), 0.1, _lib.current_stream()))
    assert g.numel() == P * stride
    for p in range(P):
        single = GeneratorPlan(gen, 1, H, W, True)
        pp = params[p * stride: p * stride + n].clone()
        y1 = single.forward(pp, x[p:p + 1].contiguous())
        g1 = single.backward(pp, dy[p:p + 1].contiguous())
        r1 = torch.zeros(gen.buffer_numel, device=DEV)
        plans1 = (C.c_void_p * 1)(single.handle)
        _lib.check(_lib.lib().splice_gen_running_stats_update(plans1, 1, _lib.ptr(r1), 0, 0.1, _lib.current_stream()))
        assert torch.equal(y[p], y1[0]), (p, (y[p] - y1[0]).abs().max().item())
        assert torch.equal(g[p * stride: p * stride + n], g1), (p, (g[p * stride: p * stride + n] - g1).abs().max().item())
        assert torch.equal(run[p], r1)
    assert not torch.equal(y[0], y[1])


def test_running_stats_match_torch_batchnorm():
    """The BatchNorm buffers after two train-mode generator calls against stock nn.BatchNorm2d (momentum 0.1, unbiased
    variance) in the same architecture (unet_general.GeneralSkip with the default arguments, parameters loaded by position:
    the module order is the reference's)."""
    from splice_amd.unet_general import GeneralSkip
    gen = GeneratorEngine(device=DEV)
    state = synth.generator_params(61, 0.02, perturb_bias=0.05)
    params = gen.flatten(state)
    ref = GeneralSkip().train()
    with torch.no_grad():
        for p_ref, (name, _) in zip(ref.parameters(), gen.table.items()):
            p_ref.copy_(torch.from_numpy(np.asarray(state[name])).reshape(p_ref.shape))
    run = torch.zeros(gen.buffer_numel, device=DEV)
    for name, (off, cnt) in gen.buffer_table.items():
        if name.endswith("running_var"):
            run[off:off + cnt] = 1.0
    plan = GeneratorPlan(gen, 1, 72, 100, False)
    for k in range(2):
        x = torch.from_numpy(synth.uniform(7, f"rs/x{k}", (1, 3, 72, 100)))
        with torch.no_grad():
            ref(x)
        plan.forward(params, x.to(DEV))
        plans = (C.c_void_p * 1)(plan.handle)
        _lib.check(_lib.lib().splice_gen_running_stats_update(plans, 1, _lib.ptr(run), 0, 0.1, _lib.current_stream()))
    ref_bufs = [b for n_, b in ref.named_buffers() if not n_.endswith("num_batches_tracked")]
    assert len(ref_bufs) == len(gen.buffer_table) == 60
    for b_ref, (name, (off, cnt)) in zip(ref_bufs, gen.buffer_table.items()):
        got = run[off:off + cnt].cpu()
        assert got.shape == b_ref.shape, name
        assert torch.allclose(got, b_ref, rtol=2e-4, atol=1e-6), (name, (got - b_ref).abs().max().item())


def _cfg(**over):
    return dict(dino_model_name="dino_vits8", dino_global_patch_size=64, **over)


def _pair_inputs(P, h, w, seed=70):
    imgs = [synth.smooth_image_pair(seed, p, h, w) for p in range(P)]
    A = torch.from_numpy(np.stack([a for a, _ in imgs])).to(DEV)
    B = torch.from_numpy(np.stack([b for _, b in imgs])).to(DEV)
    return A, B


def test_multipair_steps_bit_identical_to_single_pair_runs():
    """3 pairs on one engine == three single-pair engines, 7 steps through every regime (step 0: CLS warm-up + entire
    branch; steps >= 2: ssim + cls + id; entire branch again at step 4; unequal A/B crop sizes at step 5): per-pair losses,
    parameters, Adam moments and BatchNorm buffers bit for bit."""
    P, steps = 3, 7
    cfg = _cfg(cls_warmup=2, entire_A_every=4)
    vit_state = synth.vit_params(7, "dino_vits8", img_size=64, w_std=0.05)
    gens = [synth.generator_params(80 + p, 0.02) for p in range(P)]
    A, B = _pair_inputs(P, 64, 64)
    multi = MultiPairEngine(cfg, vit_state, gens, (64, 64), (64, 64))
    hist = []
    for i in range(steps):
        Ai = A[:, :, :60, :60].contiguous() if i == 5 else A
        multi.step(Ai, B, A)
        hist.append(multi.losses_dev.clone())
    torch.cuda.synchronize()
    for p in range(P):
        single = SpliceEngine(cfg, None, gens[p], (64, 64), (64, 64), vit_engine=multi.vit)
        for i in range(steps):
            Ai = A[p, :, :60, :60].contiguous() if i == 5 else A[p]
            single.step(Ai, B[p], A[p])
            assert torch.equal(single.losses_dev[0], hist[i][p]), (p, i, single.losses_dev[0], hist[i][p])
        torch.cuda.synchronize()
        assert torch.equal(single.params, multi.pair_params(p)), (p, (single.params - multi.pair_params(p)).abs().max().item())
        n = multi.gen.numel
        assert torch.equal(single.v, multi.v[p * multi.stride: p * multi.stride + n])
        assert torch.equal(single.running[0], multi.running[p])
        assert single.generator_calls[0] == multi.generator_calls[p] == 2 * steps + 2
    assert not torch.equal(multi.pair_params(0), multi.pair_params(1))


@pytest.mark.parametrize("P", [2, 4, 8])
def test_multipair_full_size_bit_identical_and_graph_modes(P):
    """BASELINE configs[1] shapes (224x224, ViT-B/8, T = 785): P pairs on one engine vs the single-pair runs over 4 steps
    (graph capture at the third step, replay at the fourth, on both sides), and the batched engine eager/serial vs graph/overlap.  P = 4 changes the
    GEMM tiles (128x64, 128x128) against P = 1 and keeps the merged attention backward (2 * 7 * 12 * 4 = 672 <= 1024 workgroups); P = 8 is the
    configuration of `bench.py --pairs 8` and the pairs sweep: the persistent 256x256 8-phase GEMM tile for the forward projections (and 1344
    workgroups in the merged attention backward; the two-launch form, taken above 1400, is compared bit for bit at op level:
    tests/test_ops_gpu.py::test_attention_fwd_bwd_prescaled_q_engine_forms).  None of them may change a pair's bits."""
    cfg = dict(dino_model_name="dino_vitb8", dino_global_patch_size=224)
    vit_state = synth.vit_params(7, "dino_vitb8", img_size=224, w_std=0.03)
    gens = [synth.generator_params(90 + p, 0.02) for p in range(P)]
    A, B = _pair_inputs(P, 224, 224, seed=71)
    ref = None
    vit = None
    for graph, overlap in ((1, 1), (0, 0)):
        multi = MultiPairEngine(cfg, vit_state if vit is None else None, gens, (224, 224), (224, 224), vit_engine=vit)
        vit = multi.vit
        _lib.check(_lib.lib().splice_step_use_graph(multi.handle, graph))
        _lib.check(_lib.lib().splice_step_use_overlap(multi.handle, overlap))
        for _ in range(4):
            multi.step(A, B, A)
        torch.cuda.synchronize()
        if ref is None:
            ref = (multi.params.clone(), multi.losses_dev.clone())
        else:
            assert torch.equal(multi.params, ref[0]) and torch.equal(multi.losses_dev, ref[1])
    stride, n = multi.stride, multi.gen.numel
    del multi
    for p in (0, P - 1):
        single = SpliceEngine(cfg, None, gens[p], (224, 224), (224, 224), vit_engine=vit)
        for _ in range(4):
            single.step(A[p], B[p], A[p])
        torch.cuda.synchronize()
        assert torch.equal(single.losses_dev[0], ref[1][p]), (p, single.losses_dev[0], ref[1][p])
        assert torch.equal(single.params, ref[0][p * stride: p * stride + n]), (p, (single.params - ref[0][p * stride: p * stride + n]).abs().max().item())
        del single


@pytest.mark.parametrize("mode", [True, False])
def test_top_block_modes_vs_oracle_and_batching(mode):
    """Engine option ``top_cls_only`` (default True; False = the whole top block as the reference computes it): behind the QKV projection of block 11 only the [CLS] row of every pass is computed
    (single-query attention, split-K GEMMs on one row per pass) -- all util/losses.py reads of that block besides its keys.
    Teacher-forced steps 0-2 against the fp32 oracle (same bars as the full block), and 2 pairs batched == their own
    single-pair runs bit for bit in this mode too."""
    from oracle import dino_vit
    from oracle.step import SpliceOracle
    cfg = _cfg(entire_A_every=2)
    vit_state = synth.vit_params(7, "dino_vits8", img_size=64, w_std=0.05)
    gens = [synth.generator_params(60 + p, 0.02) for p in range(2)]
    A, B = _pair_inputs(2, 64, 64, seed=72)
    eng = MultiPairEngine(cfg, vit_state, gens[:1], (64, 64), (64, 64), top_cls_only=mode)
    m = dino_vit.VisionTransformer(8, 384, 12, 6, img_size=64).eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in vit_state.items()})
    orc = SpliceOracle(m, {k: torch.from_numpy(v) for k, v in gens[0].items()}, cfg)
    for step in range(3):
        eng.params.copy_(eng.gen.flatten({k: v.detach() for k, v in orc.params.items()}))
        lo, _, og = orc.step(A[:1].cpu(), B[:1].cpu(), A[:1].cpu())
        eng.step(A[:1].contiguous(), B[:1].contiguous(), A[:1].contiguous())
        le = eng.losses(0)
        assert set(le) == set(lo)
        for k in lo:
            assert abs(le[k] - lo[k]) / abs(lo[k]) < 1e-2, (step, k, le[k], lo[k])
        num = den = 0.0
        for (name, gt), go in zip(eng.gen.unflatten(eng.grads).items(), og):
            if name.endswith("0.bias") and name != "9.0.bias":
                continue
            num += (gt.cpu().double() - go.reshape(-1).double()).norm().item() ** 2
            den += go.double().norm().item() ** 2
        assert (num / den) ** 0.5 < 3e-2, (step, (num / den) ** 0.5)
    multi = MultiPairEngine(cfg, None, gens, (64, 64), (64, 64), vit_engine=eng.vit, top_cls_only=mode)
    for _ in range(4):
        multi.step(A, B, A)
    torch.cuda.synchronize()
    for p in range(2):
        single = MultiPairEngine(cfg, None, gens[p:p + 1], (64, 64), (64, 64), vit_engine=eng.vit, top_cls_only=mode)
        for _ in range(4):
            single.step(A[p:p + 1].contiguous(), B[p:p + 1].contiguous(), A[p:p + 1].contiguous())
        torch.cuda.synchronize()
        assert torch.equal(single.params, multi.pair_params(p))
        assert torch.equal(single.losses_dev[0], multi.losses_dev[p])
