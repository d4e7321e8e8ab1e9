"""GPU parity of the generator engine (C ABI splice_gen_*) against fixtures recorded from the
reference's define_G() (forward + autograd backward) and against the fp32 oracle.

The generator runs in fp32 on the exact-f32 matrix cores, so the tolerance is fp32 rounding-order
only: outputs 2e-5 absolute (sigmoid range).  Its parameter gradients are ill-conditioned in fp32
(30 batch-1 train-mode BatchNorms in a chain): torch-CPU-fp32 itself sits 3e-3..6e-3 (relative L2)
away from the same graph evaluated in fp64, so gradients are judged against the fp64 oracle with a
2e-2 bound, and against the reference's fp32 fixtures on per-tensor energies at 3e-2.
"""
import os

import numpy as np
import pytest
import torch

from splice_amd import synth
from splice_amd.generator import GeneratorEngine, adam_step

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _stats(t):
    t = t.detach().double().cpu()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()])


@pytest.mark.parametrize("tag,h,w", [("64x64", 64, 64), ("213x213", 213, 213), ("96x130", 96, 130)])
def test_generator_vs_reference_golden(golden_dir, tag, h, w):
    g = np.load(os.path.join(golden_dir, "generator.npz"))
    eng = GeneratorEngine()
    raw = synth.generator_params(21, 0.02, perturb_bias=0.05)
    assert list(eng.table.keys()) == [str(s) for s in g["param_names"]]
    assert eng.numel == 1037523
    params = eng.flatten(raw)
    x = torch.from_numpy(synth.uniform(22, "gin/" + tag, (1, 3, h, w))).to(DEV)
    plan = eng.plan(1, h, w, need_grad=True)
    y = plan.forward(params, x)
    torch.cuda.synchronize()
    assert torch.isfinite(y).all()
    np.testing.assert_allclose(_stats(y), g[f"{tag}/out_stats"], rtol=2e-5)
    if tag == "64x64":
        np.testing.assert_allclose(y.cpu().numpy(), g[f"{tag}/out_full"], rtol=0, atol=2e-5)
    wgt = torch.from_numpy(synth.normal(23, "gw/" + tag, (1, 3, h, w))).to(DEV)
    loss = (y * wgt).sum() / y.numel() + (y * y).mean()
    np.testing.assert_allclose(loss.item(), g[f"{tag}/loss"], rtol=2e-5)
    dy = (wgt / y.numel() + 2 * y / y.numel()).contiguous()
    grads = plan.backward(params, dy)
    torch.cuda.synchronize()
    assert torch.isfinite(grads).all()
    got = np.stack([_stats(v) for v in eng.unflatten(grads).values()])
    ref = g[f"{tag}/grad_stats"]
    # |g|_1 and |g|_2^2 per tensor.  Conv biases that feed a train-mode BatchNorm have an analytically
    # ZERO gradient (BN removes the mean): both sides hold fp32 rounding noise (~1e-8) there, hence the atol.
    np.testing.assert_allclose(got[:, 1], ref[:, 1], rtol=1.5e-2, atol=3e-7)
    np.testing.assert_allclose(got[:, 2], ref[:, 2], rtol=3e-2, atol=1e-13)


@pytest.mark.parametrize("N,h,w", [(2, 75, 102), (1, 213, 213), (2, 160, 160), (1, 300, 282), (2, 264, 330)])   # (the last two: planes above 40000 / 65536 pixels --
# round 5's LDS-halo tile convolutions and weight gradients, ragged tiles in both directions, the register-resident BatchNorm segments with odd plane sizes)
def test_generator_batch_vs_fp64_oracle_and_accumulate(N, h, w):
    """N side-by-side calls == N independent oracle calls (per-call BN statistics); parameter
    gradients sum over calls; accumulate adds.  Gradients vs the fp64 oracle (see module docstring)."""
    from oracle import generator as ogen
    eng = GeneratorEngine()
    raw = synth.generator_params(5, 0.02, perturb_bias=0.03)
    params = eng.flatten(raw)
    x = torch.from_numpy(synth.uniform(6, "gx", (N, 3, h, w)))
    wgt = torch.from_numpy(synth.normal(7, "gw", (N, 3, h, w)))
    op = {k: torch.from_numpy(v).double().requires_grad_(True) for k, v in raw.items()}
    ys = torch.cat([ogen.forward(op, x[i:i + 1].double()) for i in range(N)])
    (ys * wgt.double()).sum().backward()
    plan = eng.plan(N, h, w, need_grad=True)
    y = plan.forward(params, x.to(DEV))   # the engine keeps its own copy of x
    assert (y.cpu().double() - ys.detach()).abs().max().item() < 2e-5
    grads = plan.backward(params, wgt.to(DEV))
    got = eng.unflatten(grads)
    worst, num, den = 0.0, 0.0, 0.0
    for name, p in op.items():
        if name.endswith("0.bias") and name != "9.0.bias":
            continue  # conv bias feeding a BatchNorm: analytically zero gradient, fp32 noise on both sides
        ref = p.grad.reshape(-1)
        d = (got[name].cpu().double() - ref).norm().item()
        num, den = num + d * d, den + ref.norm().item() ** 2
        err = d / (ref.norm().item() + 1e-30)
        worst = max(worst, err)
        assert err < 1e-1, (name, err)          # per tensor (4-element BN vectors are the noisiest)
    total = (num / den) ** 0.5
    print(f"    generator-grad rel err vs fp64 oracle: whole arena {total:.3e}, worst tensor {worst:.3e}")
    assert total < 1e-2, total
    g2 = plan.backward(params, wgt.to(DEV), grads=grads.clone(), accumulate=True)
    assert (g2 - 2 * grads).abs().max().item() <= 1e-5 * grads.abs().max().item() + 1e-12
    # bit-reproducible (fixed-order reductions, no float atomics)
    g3 = plan.backward(params, wgt.to(DEV))
    assert torch.equal(g3, grads)


def test_adam_matches_torch():
    n = 100003
    p0 = torch.randn(n, device=DEV)
    p = p0.clone()
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=2e-3, betas=(0.0, 0.99))
    for step in range(1, 4):
        g = torch.randn(n, device=DEV) * (10.0 ** -step)
        ref.grad = g.clone()
        opt.step()
        gg = g.clone()
        adam_step(p, gg, m, v, 2e-3, 0.0, 0.99, 1e-8, step, zero_grad=True)
        assert gg.abs().max().item() == 0.0
        assert (p - ref.data).abs().max().item() < 1e-6


def test_inversion_net_on_hip_matches_reference_golden(golden_dir):
    """SURVEY 8f rank 4 / VERDICT r1 #8: the feature-inversion generator of inversion.py:21-25 -- 6 scales, 7/7/5/5/3/3 filters,
    reflection padding (models/unet/common.py:113-118), non-RGB input -- on the HIP generator engine (5x5 / 7x7 implicit-GEMM
    instantiations, reflected gather, padded-domain data gradient + mirror fold, row-tiled weight gradient), against the
    outputs and parameter gradients recorded from the REFERENCE's models/unet/skip.py (tests/golden/inversion_net.npz, the
    fixture that also pins the CPU GeneralSkip).  96x72 and 100x84 (the second exercises Concat's centre crop on odd sizes)."""
    from oracle.fixtures import INVERSION_NET, sample, stats
    from splice_amd.networks import SkipGenerator, skip
    from splice_amd.unet_general import GeneralSkip
    g = np.load(os.path.join(golden_dir, "inversion_net.npz"))
    net = skip(8, 3, device=DEV, **INVERSION_NET)
    assert isinstance(net, SkipGenerator)                       # not the stock-PyTorch fallback
    params = list(net.named_parameters())
    assert len(params) == int(g["n_tensors"]) and sum(p.numel() for _, p in params) == int(g["n_params"])
    assert params[0][0] == "1.0.1.1.weight"                     # Conv2d behind its ReflectionPad2d: child "1" of the Sequential
    with torch.no_grad():
        for i, (name, p) in enumerate(params):
            off = 1.0 if p.dim() == 1 and name.endswith("weight") else 0.0
            p.copy_(torch.from_numpy(synth.normal(31, f"inv/p{i}", tuple(p.shape), 0.05, off)))
    for tag, (h, w) in {"96x72": (96, 72), "100x84": (100, 84)}.items():
        x = torch.from_numpy(synth.normal(32, "inv/x" + tag, (1, 8, h, w))).to(DEV)
        net.zero_grad()
        y = net(x)
        (y * y).mean().backward()
        assert y.shape == (1, 3, h, w)
        np.testing.assert_allclose(sample(y.detach().cpu(), 2053), g[f"{tag}/out_sample"], rtol=0, atol=1e-4)   # fp32 order: 7x7x32 = 1568-term sums, sigma 0.05 weights (measured 3e-5)
        np.testing.assert_allclose(stats(y.detach().cpu()), g[f"{tag}/out_stats"], rtol=1e-5)
        gs = np.stack([stats(p.grad.cpu()) for _, p in params])
        ref = g[f"{tag}/grad_stats"]
        # conv biases in front of a BatchNorm: analytically zero -- rounding noise in the reference, exact 0 here
        kinds = [k for _, _, k in net.engine.param_specs]
        live = np.array([not (k == "conv_b" and n != params[-1][0]) for (n, _), k in zip(params, kinds)])
        assert (gs[~live, 1] == 0).all()
        # per tensor: sum |g| and sum g^2 (the fixture stores no more) at 5e-2.  Tensors whose whole gradient is rounding-level
        # (sum g^2 < 1e-9: behind the train-mode BatchNorm of the 2x2 planes at the 6th scale) are excluded from THIS check by
        # that explicit rule -- they are covered element-wise by the fp64 comparison below -- instead of a bar that follows them.
        big = live & (ref[:, 1] > 1e-6) & (ref[:, 2] >= 1e-9)
        assert big.sum() >= 0.75 * live.sum()       # (83 of 104 live tensors at these sizes)
        np.testing.assert_allclose(gs[big, 1:], ref[big, 1:], rtol=5e-2)
        np.testing.assert_allclose(gs[live, 1:].sum(0), ref[live, 1:].sum(0), rtol=1e-2)
        # element-wise: every parameter gradient against the same architecture in fp64 (stock PyTorch modules, CPU) -- the fp32
        # reference itself sits 3e-3..6e-3 from fp64 on these ill-conditioned sums
        ref64 = GeneralSkip(8, 3, **INVERSION_NET).double()
        with torch.no_grad():
            for pr, (_, pm) in zip(ref64.parameters(), params):
                pr.copy_(pm.detach().cpu().double())
        y64 = ref64(x.cpu().double())
        (y64 * y64).mean().backward()
        assert (y.detach().cpu().double() - y64.detach()).abs().max().item() < 1e-4
        num = den = 0.0
        for pr, (n, pm), lv in zip(ref64.parameters(), params, live):
            if lv:
                num += (pm.grad.cpu().double() - pr.grad).norm().item() ** 2
                den += pr.grad.norm().item() ** 2
        rel = (num / den) ** 0.5
        print(f"    inversion net {tag}: whole-arena gradient rel-L2 vs fp64 = {rel:.3e}")
        assert rel < 1e-2, rel


def test_fresh_inversion_net_is_initialised_and_trainable():
    """``inversion.make_net()`` -- ``skip()`` of a non-default architecture WITHOUT init_weights, as
    inversion.py:21-25 uses it -- must carry PyTorch's constructor initialisation (the reference's modules do), not an all-zero
    arena: non-zero conv weights / BatchNorm gains, and non-zero gradients on the conv weights after one backward."""
    from splice_amd import inversion
    from splice_amd.networks import SkipGenerator
    torch.manual_seed(5)
    net = inversion.make_net(8)
    assert isinstance(net, SkipGenerator)
    kinds = {n: k for n, _, k in net.engine.param_specs}
    for n, p in net.named_parameters():
        if kinds[n] == "conv_w":
            bound = 1.0 / (p.shape[1] * p.shape[2] * p.shape[3]) ** 0.5        # kaiming_uniform(a = sqrt 5): U(-1/sqrt(fan_in), +)
            assert 0.3 * bound < p.abs().mean().item() < bound and p.abs().max().item() <= bound * (1 + 1e-6), n
        elif kinds[n] == "bn_w":
            assert (p == 1).all(), n
        elif kinds[n] == "bn_b":
            assert (p == 0).all(), n
    x = torch.randn(1, 8, 96, 96, device=DEV)                        # (6 scales: >= 65 pixels per side)
    y = net(x)
    assert y.std().item() > 1e-4                                     # not a constant colour
    (y - torch.rand_like(y)).pow(2).mean().backward()
    live = [n for n, p in net.named_parameters() if kinds[n] == "conv_w" and p.grad.abs().sum().item() > 0]
    assert len(live) == sum(1 for k in kinds.values() if k == "conv_w"), "every conv weight must receive gradient"


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(1, 224, 224), (2, 150, 200)])
def test_launch_count_forms_are_bit_neutral(size):
    """Round 4: paired convolutions, the skip BatchNorm chained into the concat BatchNorm's kernels and split-K slabs summed by the
    consuming BatchNorm (both directions) only change HOW MANY launches a pass takes -- policies that may depend on the number of
    images per launch, so they must not change a bit (the switches are read once per process: one child interpreter per setting).
    The chained kernels share their per-element arithmetic with the stand-alone ones through helpers with pinned roundings: left to
    the compiler, `-ffp-contract=fast` fused differently in the two code shapes (found with this test's tool)."""
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(__file__), "..", "tools", "gen_bits.py")
    outs = {}
    for tag, env in (("default", {}), ("no chain", {"SPLICE_BN_CHAIN": "0"}), ("no deferred slabs", {"SPLICE_BN_BWD_SLABS": "0"}),
                     ("no pairs, no chain", {"SPLICE_CONV_PAIR": "0", "SPLICE_BN_CHAIN": "0"}),
                     ("scalar weight-gradient reduce", {"SPLICE_WGRAD_REDUCE_VEC": "0"})):
        r = subprocess.run([sys.executable, tool] + [str(v) for v in size], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[tag] = [l for l in r.stdout.splitlines() if l and not l.startswith("/")]
        assert len(outs[tag]) > 100
    for tag, lines in outs.items():
        assert lines == outs["default"], (tag, [(a, b) for a, b in zip(lines, outs["default"]) if a != b][:5])


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(1, 500, 470), (2, 400, 384)])
def test_big_plane_kernels_are_bit_neutral(size):
    """Round 5 (the reference's default 855 .. 900 crops, 448^2, 512^2): on big planes the 3x3 stride-1 convolutions run a 2-D pixel tile with the
    input halo staged in LDS (conv3x3_tile_kernel) -- same k order, same chunking, same operand layout as the implicit-GEMM kernel, so where that one
    runs a single wave group without split-K (every plane of these sizes above 2048 pixel tiles) the two must agree bit for bit; the BatchNorm backward
    of big planes re-forms the activation's sign from y instead of loading the activated tensor (same comparison, same bits).  Sizes with ragged tiles in
    both directions and with two images per launch; one child interpreter per setting.  (The tile kernel's own threshold is 40000 pixels; below 2048 pixel
    tiles = 131072 pixels the implicit-GEMM kernel splits the k steps over two wave groups -- another summation order -- so the comparison raises the
    threshold to that for every arm.)"""
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(__file__), "..", "tools", "gen_bits.py")
    outs = {}
    base = dict(os.environ, SPLICE_CONV_TILE_MIN="131072")
    for tag, env in (("default", {}), ("implicit-GEMM convolutions only", {"SPLICE_CONV_TILE": "0"}), ("activation sign from the activated tensor", {"SPLICE_BN_SIGN_FROM_Y": "0"}),
                     ("one channel fragment per workgroup", {"SPLICE_CONV_BIG_FN": "0", "SPLICE_CONV_TILE": "0"})):
        r = subprocess.run([sys.executable, tool] + [str(v) for v in size], env=dict(base, **env), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[tag] = [l for l in r.stdout.splitlines() if l and not l.startswith("/")]
        assert len(outs[tag]) > 100
    for tag, lines in outs.items():
        assert lines == outs["default"], (tag, [(a, b) for a, b in zip(lines, outs["default"]) if a != b][:5])


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(1, 300, 280), (2, 256, 320)])
def test_tile_weight_gradient_matches_the_batched_kernel(size, tmp_path):
    """Round 5: the 3x3 stride-1 layers of planes above 40000 pixels take their weight gradient from conv_wgrad_tile_kernel (input patch in LDS, output
    gradient straight from global memory, accumulators in registers over a strip of tiles, four waves added in wave order) -- another summation order than
    conv_wgrad_batched_kernel's, so not the same bits: every gradient tensor within 1e-5 relative L2 of the other kernel's (measured 4e-7 .. 1.6e-6), all
    tensors the switch cannot touch identical, ragged tiles in both directions and two images per launch.  (Both against the fp32 oracle: the step tests.)"""
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(__file__), "..", "tools", "gen_bits.py")
    dumps = {}
    for tag, env in (("tile", {"SPLICE_WGRAD_TILE": "1"}), ("batched", {"SPLICE_WGRAD_TILE": "0"})):
        path = str(tmp_path / f"{tag}.pt")
        r = subprocess.run([sys.executable, tool] + [str(v) for v in size], env=dict(os.environ, GEN_BITS_DUMP=path, **env), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        dumps[tag] = torch.load(path)
    differing, worst = 0, 0.0
    for k, a in dumps["tile"].items():
        b = dumps["batched"][k]
        rel = (a.double() - b.double()).norm().item() / max(b.double().norm().item(), 1e-30)
        differing += rel > 0
        worst = max(worst, rel)
        assert rel < 1e-5, (k, rel)
    print(f"    tile weight gradient vs batched kernel at {size}: {differing} tensors differ, worst relative L2 {worst:.2e}")
    assert 1 <= differing <= 8, differing   # only the weights of the 3x3 stride-1 layers of the big planes
