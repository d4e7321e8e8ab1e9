"""GPU: the fp8 operand path of BASELINE configs[4] ("fp8 MFMA attention + self-sim path"): e4m3 (OCP fp8, the gfx950 MFMA
format) operands for the QKV / fc1 / fc2 forward projections (block-scaled K = 128 MFMA with unit block scales) and for the key
self-similarity Gram matrices.

Two layers of checks:
  * EXACTNESS of the machinery: the quantisers and the fp8 GEMM against a torch emulation that rounds through
    torch.float8_e4m3fn -- products of e4m3 values are exact in fp32, so the only freedom is the fp32 summation order
    (5e-5 relative against an fp64 sum).  This pins the number format (e4m3fn, not the MI300 fnuz variant) and the k-permutation of the tile
    engine's fp8 mode.
  * ACCURACY of the path: its own tolerance table against the fp32 oracle (the bf16 path's bars do not apply: e4m3 keeps 3
    mantissa bits), step losses / gradients teacher-forced as in tests/test_step_gpu.py.  Bars are set from the measured
    values printed by the test."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from splice_amd import _lib, synth
from splice_amd.engine import SpliceEngine

pytestmark = pytest.mark.gpu
DEV = "cuda"
E4M3_MAX = 448.0


def _quant_ref(x):
    """row-wise e4m3 quantisation as the kernels do it: q = e4m3(x * 448 / amax), scale = amax / 448"""
    amax = x.abs().amax(dim=1, keepdim=True).clamp_min(1e-20)
    q = (x * (E4M3_MAX / amax)).to(torch.float8_e4m3fn)
    return q, (amax / E4M3_MAX).squeeze(1)


def test_quantize_rows_matches_e4m3fn():
    L = _lib.lib()
    x = torch.from_numpy(synth.normal(3, "fp8/q", (37, 768), 1.7)).to(DEV)
    x[5] = 0.0
    x[7, 3] = 300.0
    q = torch.zeros(37, 768, dtype=torch.uint8, device=DEV)
    sc = torch.zeros(37, device=DEV)
    _lib.check(L.splice_quantize_rows_fp8(_lib.ptr(x), 768, _lib.ptr(q), 768, _lib.ptr(sc), 37, 768, _lib.current_stream()))
    qr, sr = _quant_ref(x)
    assert torch.allclose(sc, sr, rtol=1e-6, atol=0)
    got = q.view(torch.float8_e4m3fn).float()
    want = qr.float()
    # round-to-nearest-even on both sides; a product x * (448 / amax) that lands within one ulp of a tie may round either way
    mism = (got != want).float().mean().item()
    assert mism < 2e-3, mism
    assert (got - want).abs().max().item() <= 32.0            # at most one e4m3 step at the top binade
    assert got.abs().max().item() == E4M3_MAX and (got[5] == 0).all()


@pytest.mark.parametrize("M,N,K", [(800, 2304, 768), (1600, 2304, 768), (3200, 1152, 384), (130, 72, 128)])
def test_fp8_gemm_exact_vs_emulation(M, N, K):
    """C = (qA * sA) (qB * sB)^T + bias on the fp8 MFMA == the same product formed in fp32 from the e4m3 values."""
    L = _lib.lib()
    A = torch.from_numpy(synth.normal(4, f"fp8/A{M}", (M, K), 1.0)).to(DEV)
    B = torch.from_numpy(synth.normal(5, f"fp8/B{N}", (N, K), 0.03)).to(DEV)
    bias = torch.from_numpy(synth.normal(6, f"fp8/b{N}", (N,), 0.1)).to(DEV)
    qa, sa = _quant_ref(A)
    qb, sb = _quant_ref(B)
    out = torch.zeros(M, N, device=DEV, dtype=torch.bfloat16)
    outT = torch.zeros(N, (M + 3) // 4 * 4, device=DEV, dtype=torch.bfloat16)
    f32 = torch.zeros(M, N, device=DEV)
    e = _lib.GemmEpilogue()
    e.bias = bias.data_ptr(); e.out_bf = out.data_ptr(); e.ldbf = N; e.out_bf_t = outT.data_ptr(); e.ldt = outT.shape[1]
    e.out_f32_cols = f32.data_ptr(); e.ld_cols = N; e.col_lo = 0; e.col_hi = N
    e.row_scale = sa.data_ptr(); e.col_scale = sb.data_ptr()
    fl = _lib.EPI_SCALE_RC | _lib.EPI_BIAS | _lib.EPI_OUT_BF | _lib.EPI_OUT_T | _lib.EPI_COLS_F32
    qa8, qb8 = qa.view(torch.uint8).contiguous(), qb.view(torch.uint8).contiguous()
    _lib.check(L.splice_gemm_nt_fp8(fl, _lib.ptr(qa8), K, _lib.ptr(qb8), K, M, N, K, C.byref(e), _lib.current_stream()))
    torch.cuda.synchronize()
    ref = ((qa.double() @ qb.double().T) * sa.double()[:, None] * sb.double()[None, :] + bias.double()).float()
    err = ((f32 - ref).norm() / ref.norm()).item()
    assert err < 5e-5, err      # fp32 accumulation of 768 exact products (a wrong format or k-permutation would be O(1))
    assert ((out.float() - ref).norm() / ref.norm()).item() < 4e-3          # bf16 output rounding
    assert torch.equal(outT[:, :M], out.T)
    # and the quantisation error itself, for the record: fp8 product vs the fp32 product of the unquantised operands
    true = A @ B.T + bias
    print(f"    fp8 GEMM {M}x{N}x{K}: summation-order err {err:.1e}; quantisation err vs fp32 operands {((f32 - true).norm() / true.norm()).item():.3e}")


def test_fp8_vit_features_vs_oracle():
    """ViT-B/8 layer-11 keys with fp8 QKV / fc1 / fc2 projections in all 12 layers against the fp32 oracle ViT (bf16 path: 2e-2), 224x224."""
    from oracle import dino_vit
    from oracle import extractor as oext
    from splice_amd.vit import KIND_QKV_LAST_F32, VitEngine
    name, size = "dino_vitb8", 224
    vit_state = synth.vit_params(7, name, img_size=size, w_std=0.03)
    img = torch.from_numpy(synth.normal(13, "fp8/img", (1, 3, size, size)))
    m = dino_vit.VisionTransformer(8, 768, 12, 12, img_size=size).eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in vit_state.items()})
    with torch.no_grad():
        k_ref = oext.keys_from_input(m, img, 11)                  # [h, T, d]
    rel = {}
    for fp8 in (False, True):
        eng = VitEngine(name, device=DEV).load_state_dict(vit_state)
        if fp8:
            eng.enable_fp8("attention")
        ctx = eng.context(1, size, size, need_grad=False)
        ctx.forward(img.to(DEV), normalize=False)
        qkv = ctx.read(KIND_QKV_LAST_F32, 11)[0, : ctx.T].cpu()
        k_got = qkv[:, 768:1536].reshape(ctx.T, 12, 64).permute(1, 0, 2)
        rel[fp8] = ((k_got.double() - k_ref.double()).norm() / k_ref.double().norm()).item()
    print(f"    layer-11 keys rel-L2 vs fp32 oracle: bf16 path {rel[False]:.3e}, fp8 path {rel[True]:.3e}")
    assert rel[False] < 2e-2 and rel[True] < FP8_KEYS_TOL, rel


# ---- the fp8 path's own tolerance table: bars <= 1.3x the worst value measured in round 3 (printed by the tests)
# measured r3, QKV + fc1 + fc2 forward projections, the attention forward and the Gram matrices in e4m3 (worst over steps 0-2;
# without the fp8 attention: cls 1.84e-2, gradient 1.54e-1): layer-11 keys 9.2e-2 (an
# e4m3 GEMM of uncorrelated operands carries ~3.7e-2 relative error by itself, test_fp8_gemm_exact_vs_emulation prints it; three
# of them per block accumulate through the residual stream); losses <= 2.53e-2 (cls / total), <= 2.56e-2 (ssim), <= 1.2e-2 (id);
# generator gradient 1.65e-1 (ViT-S/8 @ 64) / 9.8e-2 (ViT-B/8 @ 224) / 1.7e-1 (configs[4] ssim term at 224 + 320 + 448) -- against
# 6e-3 / 3e-3 / 7e-3 on the bf16 path.  What the optimisation makes of it: test_fp8_trajectory_reaches_reference_level.
FP8_KEYS_TOL = 1.2e-1
FP8_LOSS_TOL = {"loss": 3.3e-2, "loss_global_cls": 3.3e-2, "loss_entire_cls": 3.3e-2, "loss_global_ssim": 3.3e-2, "loss_entire_ssim": 3.3e-2, "loss_global_id_B": 1.6e-2}
FP8_GRAD_TOL = 2.1e-1


@pytest.mark.parametrize("name,size", [("dino_vits8", 64), ("dino_vitb8", 224)])
def test_fp8_step_vs_oracle_tolerance_table(name, size):
    from oracle import dino_vit
    from oracle.step import SpliceOracle
    cfg = dict(dino_model_name=name, dino_global_patch_size=size)
    A, B = synth.smooth_image_pair(123, 0, size, size)
    vit_state = synth.vit_params(7, name, img_size=size, w_std=0.05 if size == 64 else 0.03)
    gen_state = synth.generator_params(9, 0.02)
    eng = SpliceEngine(cfg, vit_state, gen_state, (size, size), (size, size), fp8="attention")
    patch, dim, depth, heads = dino_vit.DINO_CONFIGS[name]
    m = dino_vit.VisionTransformer(patch, dim, depth, heads, img_size=size).eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in vit_state.items()})
    orc = SpliceOracle(m, {k: torch.from_numpy(v) for k, v in gen_state.items()}, cfg)
    At, Bt = torch.from_numpy(A), torch.from_numpy(B)
    worst = {}
    for step in range(3):
        eng.params.copy_(eng.gen.flatten({k: v.detach() for k, v in orc.params.items()}))
        lo, _, og = orc.step(At[None], Bt[None], At[None])
        eng.step(At.to(DEV), Bt.to(DEV), At.to(DEV))
        le = eng.losses()
        assert set(le) == set(lo)
        for k in lo:
            r = abs(le[k] - lo[k]) / abs(lo[k])
            worst[k] = max(worst.get(k, 0.0), r)
        num = den = 0.0
        for (pname, gt), go in zip(eng.gen.unflatten(eng.grads).items(), og):
            if pname.endswith("0.bias") and pname != "9.0.bias":
                continue
            num += (gt.cpu().double() - go.reshape(-1).double()).norm().item() ** 2
            den += go.double().norm().item() ** 2
        worst["grad"] = max(worst.get("grad", 0.0), (num / den) ** 0.5)
    print(f"    fp8 path {name}@{size}: worst relative deviations over steps 0-2 vs fp32 oracle: " + ", ".join(f"{k} {v:.3e}" for k, v in worst.items()))
    for k, v in worst.items():
        assert v < (FP8_GRAD_TOL if k == "grad" else FP8_LOSS_TOL[k]), (k, v)


# ---- BASELINE configs[4] at its own sizes: every loss term at the ViT input scales 224 / 320 / 448 with the fp8 operand path
@pytest.mark.parametrize("term", ["cls", "ssim", "id"])
def test_configs4_multiscale_224_320_448_fp8_vs_oracle(term):
    """``MultiScaleEngine(scales=(224, 320, 448), fp8="attention")`` on ViT-B/8 (T = 785 / 1601 / 3137; bilinear Resize 224 -> 320 / 448 and
    its adjoint, interpolated position tables, 32-query attention waves and the two-launch attention backward at the large
    scales, fp8 QKV / fc1 / fc2 + fp8 self-similarity Gram): summed loss and whole-arena generator gradient of one step against
    the fp32 CPU oracle evaluated at the three ``dino_global_patch_size`` values on the same generator outputs.  One loss term per
    case (the other lambdas are zero on both sides) so that the oracle's autograd holds one differentiated ViT pass per scale."""
    from oracle import losses as OL
    from oracle import dino_vit
    from oracle.step import SpliceOracle
    from splice_amd.engine import MultiScaleEngine
    scales = (224, 320, 448)
    lam = dict(lambda_global_cls=0.0, lambda_global_ssim=0.0, lambda_global_identity=0.0, lambda_entire_cls=0.0, lambda_entire_ssim=0.0)
    lam[{"cls": "lambda_global_cls", "ssim": "lambda_global_ssim", "id": "lambda_global_identity"}[term]] = {"cls": 10.0, "ssim": 1.0, "id": 1.0}[term]
    cfg = dict(dino_model_name="dino_vitb8", entire_A_every=10 ** 9, cls_warmup=0, **lam)
    A, B = synth.smooth_image_pair(224, 3, 224, 224)
    A, B = torch.from_numpy(A), torch.from_numpy(B)
    vit_state = synth.vit_params(7, "dino_vitb8", img_size=224, w_std=0.03)
    gen_state = synth.generator_params(9, 0.02)
    eng = MultiScaleEngine(cfg, vit_state, gen_state, (224, 224), None, scales=scales, fp8="attention")
    assert [e.ctx_g.T for e in eng.engines] == [785, 1601, 3137] and all(e.ctx_g.fp8 for e in eng.engines)
    m = dino_vit.VisionTransformer(8, 768, 12, 12, img_size=224).eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in vit_state.items()})
    orcs = [SpliceOracle(m, {k: torch.from_numpy(v) for k, v in gen_state.items()}, dict(cfg, dino_global_patch_size=sz)) for sz in scales]
    inputs = {"step": 0, "A_global": A[None], "B_global": B[None]}
    outputs = orcs[0].model_forward(inputs)
    og = [torch.zeros_like(p) for p in orcs[0].params.values()]
    per = []
    for o in orcs:                      # one scale at a time: its graph is freed before the next one is built
        o.params = orcs[0].params
        d = OL.loss_g(o.vit, o.cfg, o.lambdas, outputs, inputs)
        g = torch.autograd.grad(d["loss"], list(orcs[0].params.values()), allow_unused=True, retain_graph=True)
        for acc, gi in zip(og, g):
            if gi is not None:
                acc += gi
        per.append({k: float(v.detach()) for k, v in d.items()})
        del d, g
    total = sum(d["loss"] for d in per)
    eng.step(A.to(DEV), B.to(DEV), None)
    le = eng.losses()
    worst = abs(le["loss"] - total) / abs(total)
    for sz, d in zip(scales, per):
        for k, v in d.items():
            worst = max(worst, abs(le["scales"][sz][k] - v) / abs(v))
    num = den = 0.0
    for (pname, gt), go in zip(eng.gen.unflatten(eng.grads).items(), og):
        if pname.endswith("0.bias") and pname != "9.0.bias":
            continue
        num += (gt.cpu().double() - go.reshape(-1).double()).norm().item() ** 2
        den += go.double().norm().item() ** 2
    rel = (num / den) ** 0.5
    print(f"    configs[4] {term}: summed loss {le['loss']:.4f} vs oracle {total:.4f} (worst per-scale term deviation {worst:.3e}); gradient rel err {rel:.3e}")
    assert worst < FP8_LOSS_TOL["loss_global_ssim" if term == "ssim" else "loss"], worst
    assert rel < FP8_GRAD_TOL, rel


def test_fp8_trajectory_reaches_reference_level(golden_dir):
    """VERDICT r2 #3d: the 78-step fixture recorded from the REFERENCE loop (tests/golden/steps.npz, 64x64 pair, ViT-S/8-shaped
    stand-in) optimised with the fp8 operand path: steps 0-2 within the fp8 loss table, and the level reached over steps
    60..74 not worse than 1.25x the reference's -- the bar the bf16 path is held to (tests/test_step_gpu.py)."""
    g = np.load(os.path.join(golden_dir, "steps.npz"))
    A, B = synth.smooth_image_pair(32, 0, 64, 64)
    cfg = dict(dino_model_name="dino_vits8", dino_global_patch_size=64)
    eng = SpliceEngine(cfg, synth.vit_params(7, "dino_vits8", img_size=64, w_std=0.05), synth.generator_params(31, 0.02), (64, 64), (64, 64), fp8="attention")
    At, Bt = torch.from_numpy(A).to(DEV), torch.from_numpy(B).to(DEV)
    mine = []
    for _ in range(78):
        eng.step(At, Bt, At)
        mine.append(eng.losses()["loss"])
    mine, ref = np.array(mine), g["a/losses"][:, 0]
    assert np.isfinite(mine).all()
    for i in range(3):
        assert abs(mine[i] - ref[i]) / ref[i] < 6.2e-2, (i, mine[i], ref[i])     # measured 4.8e-2 at step 0 (the fp8 attention's share: 1.9e-2 without it)
    tail_mine, tail_ref = np.sort(mine[60:75])[:5].mean(), np.sort(ref[60:75])[:5].mean()
    print(f"    fp8 path, level reached (steps 60..74): {tail_mine:.1f} vs reference {tail_ref:.1f}; first steps {mine[:3]} vs {ref[:3]}")
    assert tail_mine < 1.25 * tail_ref


def test_fp8_is_a_property_of_the_context_not_of_the_shared_vit():
    """engines that share one frozen ViT keep their own precision -- a bf16 engine built AFTER an fp8 engine on the
    same VitEngine produces bit for bit what it produces on a ViT no fp8 engine ever touched."""
    cfg = dict(dino_model_name="dino_vits8", dino_global_patch_size=64)
    vit_state = synth.vit_params(7, "dino_vits8", img_size=64, w_std=0.05)
    gen_state = synth.generator_params(9, 0.02)
    A, B = synth.smooth_image_pair(123, 0, 64, 64)
    At, Bt = torch.from_numpy(A).to(DEV), torch.from_numpy(B).to(DEV)

    def run(eng):
        for _ in range(3):
            eng.step(At, Bt, At)
        torch.cuda.synchronize()
        return eng.params.clone()

    clean = run(SpliceEngine(cfg, vit_state, gen_state, (64, 64), (64, 64)))
    e8 = SpliceEngine(cfg, vit_state, gen_state, (64, 64), (64, 64), fp8="attention")
    p8 = run(e8)
    shared = run(SpliceEngine(cfg, None, gen_state, (64, 64), (64, 64), vit_engine=e8.vit))
    assert e8.ctx_g.fp8 and not torch.equal(p8, clean)
    assert torch.equal(shared, clean)


def _e4m3(x):
    return x.clamp(-448, 448).to(torch.float8_e4m3fn)


@pytest.mark.parametrize("B,T,variant", [(2, 785, 0), (1, 785, 12), (3, 197, 1), (1, 1601, 0)])
def test_fp8_attention_forward(B, T, variant):
    """splice_attention_fwd_fp8 (Q K^T and P V on the fp8 MFMA from unscaled e4m3 q, k, v; e4m3 probabilities against a
    reference point 2^-6 below the running maximum) against fp32 attention on the SAME e4m3-rounded operands -- isolates what the
    kernel adds: the quantisation of P (measured 2.3e-2 rel-L2 on N(0, 1) operands; bar 3e-2) -- and, for the record, against fp32
    attention on the unquantised operands.  Rows of padding tokens are masked keys; variants: automatic / 32 queries per wave in
    two key groups / 16 queries in one group (the launch forms must agree bit for bit, as in the bf16 kernel)."""
    D, H = 768, 12
    Tld = (T + 31) // 32 * 32
    rows = B * Tld
    g = torch.Generator().manual_seed(11)
    qkv = torch.randn(rows, 3 * D, generator=g)
    qkv[:, D:2 * D] += 0.5                      # key bias: scores with a common offset (softmax is shift invariant)
    q8 = _e4m3(qkv).to(DEV)
    q8T = q8.view(torch.uint8).T.contiguous()
    deq = q8.float().reshape(B, Tld, 3, H, 64)[:, :T]

    def ref(x):
        q, k, v = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)
        s = (q @ k.transpose(-1, -2)) * 0.125
        return (s.softmax(-1) @ v).transpose(1, 2).reshape(B, T, D), torch.logsumexp(s, -1)

    want, lse_want = ref(deq)
    true, _ = ref(qkv.to(DEV).reshape(B, Tld, 3, H, 64)[:, :T])
    L = _lib.lib()
    outs = []
    for v in ([variant] if variant else [0, 1, 11, 2]):
        L.splice_attention_variant(v)
        out = torch.zeros(rows, D, device=DEV, dtype=torch.bfloat16)
        lse = torch.zeros(B, H, Tld, device=DEV)
        _lib.check(L.splice_attention_fwd_fp8(_lib.ptr(q8.view(torch.uint8)), _lib.ptr(q8T), rows, B, T, Tld, D, H, 0.125, _lib.ptr(out), _lib.ptr(lse),
                                              _lib.current_stream()))
        torch.cuda.synchronize()
        outs.append(out.clone())
    L.splice_attention_variant(0)
    got = outs[0].float().reshape(B, Tld, D)[:, :T]
    e_same = ((got - want).norm() / want.norm()).item()
    e_true = ((got - true).norm() / true.norm()).item()
    e_lse = (lse[:, :, :T] * 0.6931471805599453 - lse_want).abs().max().item()
    print(f"    fp8 attention B{B} T{T}: rel-L2 vs fp32 attention on the e4m3 operands {e_same:.3e}, vs unquantised operands {e_true:.3e}; lse abs err {e_lse:.2e}")
    assert e_same < 3e-2 and e_lse < 2e-3
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


def test_fp8_gemm_transposed_e4m3_outputs():
    """The QKV projection of the fp8 mode writes, beside its bf16 outputs, the unscaled e4m3 copies the fp8 attention reads:
    row-major [M][N] and transposed [N][M] -- both must be the e4m3 rounding of the bf16-unrounded result."""
    M, N, K = 800, 2304, 768
    A = torch.from_numpy(synth.normal(4, "fp8t/A", (M, K), 1.0)).to(DEV)
    Bm = torch.from_numpy(synth.normal(5, "fp8t/B", (N, K), 0.03)).to(DEV)
    bias = torch.from_numpy(synth.normal(6, "fp8t/b", (N,), 0.1)).to(DEV)
    qa, sa = _quant_ref(A)
    qb, sb = _quant_ref(Bm)
    out = torch.zeros(M, N, device=DEV, dtype=torch.bfloat16)
    outT = torch.zeros(N, M, device=DEV, dtype=torch.bfloat16)
    o8 = torch.zeros(M, N, device=DEV, dtype=torch.uint8)
    o8T = torch.zeros(N, M, device=DEV, dtype=torch.uint8)
    e = _lib.GemmEpilogue()
    e.bias = bias.data_ptr(); e.out_bf = out.data_ptr(); e.ldbf = N; e.out_bf_t = outT.data_ptr(); e.ldt = M
    e.row_scale = sa.data_ptr(); e.col_scale = sb.data_ptr()
    e.out_f8 = o8.data_ptr(); e.ld8 = N; e.out_f8_t = o8T.data_ptr(); e.ldt8 = M
    fl = _lib.EPI_SCALE_RC | _lib.EPI_BIAS | _lib.EPI_OUT_BF | _lib.EPI_OUT_T | _lib.EPI_OUT_F8 | _lib.EPI_OUT_F8T
    _lib.check(_lib.lib().splice_gemm_nt_fp8(fl, _lib.ptr(qa.view(torch.uint8).contiguous()), K, _lib.ptr(qb.view(torch.uint8).contiguous()), K, M, N, K,
                                              C.byref(e), _lib.current_stream()))
    torch.cuda.synchronize()
    ref = ((qa.double() @ qb.double().T) * sa.double()[:, None] * sb.double()[None, :] + bias.double()).float()
    got8 = o8.view(torch.float8_e4m3fn).float()
    want8 = _e4m3(ref).float()
    # summation order moves a value across an e4m3 rounding boundary now and then: at most one ulp (2^-3 relative), rarely
    assert ((got8 - want8).abs() <= want8.abs() * 0.126 + 2.1e-3).all()      # (2^-9 = e4m3's subnormal spacing)
    assert (got8 != want8).float().mean().item() < 2e-3
    assert torch.equal(o8T, o8.T)
    assert torch.equal(outT, out.T)
