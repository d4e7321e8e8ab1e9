"""GPU: BASELINE configs[4] groundwork -- several global crops per image (the reference's global_*_crops_n_crops > 1:
netG sees the crops as ONE batch, every loss term is summed over the crops) and several ViT input scales per step
(an extension: the reference has one dino_global_patch_size), each against the fp32 CPU oracle at identical parameters."""
import numpy as np
import pytest
import torch

from splice_amd import synth
from splice_amd.engine import MultiScaleEngine, SpliceEngine
from splice_amd.generator import GeneratorEngine, GeneratorPlan

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("N,hw", [(3, (64, 64)), (2, (96, 130)), (4, (213, 213))])
def test_generator_batch_statistics_vs_oracle(N, hw):
    """netG on a batch of N images = nn.BatchNorm2d statistics over the whole batch (models/unet/common.py:95-96 with
    n_crops > 1): output and every parameter gradient against the functional oracle in fp64 (F.batch_norm on the batch)."""
    from oracle import generator as OG
    H, W = hw
    gen = GeneratorEngine(device=DEV)
    state = synth.generator_params(21, 0.02, perturb_bias=0.05)
    params = gen.flatten(state)
    x = synth.uniform(22, f"bs/x{N}", (N, 3, H, W))
    wgt = synth.normal(23, f"bs/w{N}", (N, 3, H, W))
    plan = GeneratorPlan(gen, N, H, W, True, batch_stats=True)
    y = plan.forward(params, torch.from_numpy(x).to(DEV))
    # loss = sum(y * w) / numel + mean(y^2)  ->  dy = w / numel + 2 y / numel
    dy = (torch.from_numpy(wgt).to(DEV) + 2.0 * y) / y.numel()
    g = plan.backward(params, dy.contiguous())
    p64 = {k: torch.from_numpy(np.asarray(v)).double().requires_grad_(True) for k, v in state.items()}
    y64 = OG.forward(p64, torch.from_numpy(x).double())
    loss = (y64 * torch.from_numpy(wgt).double()).sum() / y64.numel() + (y64 * y64).mean()
    grads = torch.autograd.grad(loss, list(p64.values()))
    assert (y.cpu().double() - y64.detach()).abs().max().item() < 2e-5
    num = den = 0.0
    for (name, gt), go in zip(gen.unflatten(g).items(), grads):
        if name.endswith("0.bias") and name != "9.0.bias":
            continue   # conv biases in front of a BatchNorm: analytically zero
        num += (gt.cpu().double() - go.reshape(-1)).norm().item() ** 2
        den += go.norm().item() ** 2
    rel = (num / den) ** 0.5
    print(f"    batch-stat generator N={N} {H}x{W}: gradient rel err vs fp64 oracle {rel:.3e}")
    assert rel < 1e-2, rel
    # and it is NOT the per-image result
    y1 = GeneratorPlan(gen, N, H, W, False).forward(params, torch.from_numpy(x).to(DEV))
    assert (y1 - y).abs().max().item() > 1e-4


def _rel_grad(eng, og):
    num = den = 0.0
    for (name, gt), go in zip(eng.gen.unflatten(eng.grads).items(), og):
        if name.endswith("0.bias") and name != "9.0.bias":
            continue
        num += (gt.cpu().double() - go.reshape(-1).double()).norm().item() ** 2
        den += go.double().norm().item() ** 2
    return (num / den) ** 0.5


def _oracle(cfg, vit_state, gen_state, img_size=64):
    from oracle import dino_vit
    from oracle.step import SpliceOracle
    m = dino_vit.VisionTransformer(8, 384, 12, 6, img_size=img_size).eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in vit_state.items()})
    return SpliceOracle(m, {k: torch.from_numpy(v) for k, v in gen_state.items()}, cfg)


@pytest.mark.parametrize("n_crops", [2, 3, (3, 2), (1, 3)])
def test_n_crops_step_vs_oracle(n_crops):
    """global_A_crops_n_crops = global_B_crops_n_crops = n (conf/default/config.yaml): A_global / B_global are [n,3,s,s] stacks
    (data/transforms.py:27), netG normalises over the stack, util/losses.py:74-105 sums each term over the crops, the entire
    branch stays one image and takes its [CLS] target from the FIRST B crop (zip).  Teacher-forced steps 0..3 (entire +
    cls; ssim + cls + id; ...; entire again at step 3) against the oracle: losses 1e-2, gradient 6e-2 (measured r5: 2.3e-3 / 4.1e-2 -- batch statistics over the crops make the gradient the noisier one)."""
    cfg = dict(dino_model_name="dino_vits8", dino_global_patch_size=64, entire_A_every=3)
    vit_state = synth.vit_params(7, "dino_vits8", img_size=64, w_std=0.05)
    gen_state = synth.generator_params(41, 0.02)
    A, B = synth.smooth_image_pair(44, 0, 72, 80)
    A, B = torch.from_numpy(A), torch.from_numpy(B)
    # (nA, nB) unequal: the reference zips the crop lists (util/losses.py:76,87,98) -- structure term over the nA crops, identity
    # term over the nB crops, appearance term over min(nA, nB) pairs -- while netG's BatchNorm sees all crops of its call
    nA, nB = (n_crops, n_crops) if isinstance(n_crops, int) else n_crops
    offs = [(0, 0), (8, 16), (4, 9)][:nA]
    Ac = torch.stack([A[:, t:t + 64, l:l + 64] for t, l in offs]).contiguous()
    Bc = torch.stack([B[:, t:t + 64, l:l + 64] for t, l in [(8, 0), (0, 16), (3, 7)][:nB]]).contiguous()
    eng = SpliceEngine(cfg, vit_state, gen_state, (64, 64), (72, 80), n_crops=n_crops)
    orc = _oracle(cfg, vit_state, gen_state)
    for step in range(4):
        eng.params.copy_(eng.gen.flatten({k: v.detach() for k, v in orc.params.items()}))
        lo, _, og = orc.step(Ac, Bc, A[None])
        eng.step(Ac.to(DEV), Bc.to(DEV), A.to(DEV))
        le = eng.losses()
        assert set(le) == set(lo), (step, sorted(le), sorted(lo))
        for k in lo:
            assert abs(le[k] - lo[k]) / abs(lo[k]) < 1e-2, (step, k, le[k], lo[k])
        rel = _rel_grad(eng, og)
        print(f"    n_crops={n_crops} step {step}: loss {le['loss']:.4f} vs oracle {lo['loss']:.4f}; gradient rel err {rel:.3e}")
        assert rel < 6e-2, (step, rel)


def test_multiscale_step_vs_oracle():
    """The same crops seen at two ViT input scales (64: identity Resize; 96: bilinear 64 -> 96, interpolated position table),
    loss = sum over the scales of the reference loss; one Adam step on the summed gradient.  Teacher-forced steps 0..2
    against the oracle evaluated at both dino_global_patch_size values on the same generator outputs."""
    from oracle import losses as OL
    scales = (64, 96)
    cfg = dict(dino_model_name="dino_vits8", entire_A_every=2)
    vit_state = synth.vit_params(7, "dino_vits8", img_size=64, w_std=0.05)
    gen_state = synth.generator_params(43, 0.02)
    A, B = synth.smooth_image_pair(45, 1, 64, 64)
    A, B = torch.from_numpy(A), torch.from_numpy(B)
    eng = MultiScaleEngine(cfg, vit_state, gen_state, (64, 64), (64, 64), scales=scales)
    orcs = [_oracle(dict(cfg, dino_global_patch_size=sz), vit_state, gen_state) for sz in scales]
    for o in orcs[1:]:
        o.params = orcs[0].params
    for step in range(3):
        eng.params.copy_(eng.gen.flatten({k: v.detach() for k, v in orcs[0].params.items()}))
        inputs = {"step": step, "A_global": A[None], "B_global": B[None]}
        if step % 2 == 0:
            inputs["A"] = A[None]
        outputs = orcs[0].model_forward(inputs)
        per = [OL.loss_g(o.vit, o.cfg, o.lambdas, outputs, inputs) for o in orcs]
        total = sum(d["loss"] for d in per)
        og = torch.autograd.grad(total, list(orcs[0].params.values()), allow_unused=True)
        og = [torch.zeros_like(p) if g is None else g for g, p in zip(og, orcs[0].params.values())]
        eng.step(A.to(DEV), B.to(DEV), A.to(DEV))
        le = eng.losses()
        total_f = float(total.detach())
        assert abs(le["loss"] - total_f) / abs(total_f) < 1e-2, (step, le["loss"], total_f)
        for sz, d in zip(scales, per):
            for k, v in d.items():
                assert abs(le["scales"][sz][k] - float(v.detach())) / abs(float(v.detach())) < 1e-2, (step, sz, k)
        rel = _rel_grad(eng, og)
        print(f"    multi-scale step {step}: loss {le['loss']:.4f} vs oracle {total_f:.4f}; gradient rel err {rel:.3e}")
        assert rel < 3e-2, (step, rel)
        orcs[0].opt.step(og)
    # the update itself: parameters after the last fused Adam against the oracle's Adam on the oracle gradient
    ref = eng.gen.flatten({k: v.detach() for k, v in orcs[0].params.items()})
    # beta1 = 0: an update is lr * g / sqrt(v_hat) <= 1.72 lr at t = 3, so an element whose tiny gradient differs in sign moves
    # by up to 2 x 1.72 x lr = 6.9e-3; those must be rare -- everything else agrees to rounding
    diff = (eng.params - ref).abs()
    assert diff.max().item() < 8e-3 and diff.mean().item() < 5e-5 and (diff > 1e-4).float().mean().item() < 0.02, \
        (diff.max().item(), diff.mean().item(), (diff > 1e-4).float().mean().item())
