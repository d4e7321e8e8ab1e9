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


@pytest.mark.parametrize("N,h,w", [(2, 75, 102), (1, 213, 213), (2, 160, 160)])
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
