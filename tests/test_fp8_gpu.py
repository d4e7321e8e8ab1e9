"""GPU: the fp8 operand path of BASELINE configs[4] ("fp8 MFMA attention + self-sim path"): e4m3 (OCP fp8, the gfx950 MFMA
format) operands on v_mfma_f32_16x16x32_fp8_fp8 for the QKV projection and for the key self-similarity Gram matrices.

Two layers of checks:
  * EXACTNESS of the machinery: the quantisers and the fp8 GEMM against a torch emulation that rounds through
    torch.float8_e4m3fn -- products of e4m3 values are exact in fp32, so the only freedom is the fp32 summation order
    (5e-5 relative against an fp64 sum).  This pins the number format (e4m3fn, not the MI300 fnuz variant) and the k-permutation of the tile
    engine's fp8 mode.
  * ACCURACY of the path: its own tolerance table against the fp32 oracle (the bf16 path's bars do not apply: e4m3 keeps 3
    mantissa bits), step losses / gradients teacher-forced as in tests/test_step_gpu.py.  Bars are set from the measured
    values printed by the test."""
import ctypes as C

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
    """ViT-B/8 layer-11 keys with fp8 QKV projections in all 12 layers against the fp32 oracle ViT (bf16 path: 2e-2), 224x224."""
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
            eng.enable_fp8()
        ctx = eng.context(1, size, size, need_grad=False)
        ctx.forward(img.to(DEV), normalize=False)
        qkv = ctx.read(KIND_QKV_LAST_F32, 11)[0, : ctx.T].cpu()
        k_got = qkv[:, 768:1536].reshape(ctx.T, 12, 64).permute(1, 0, 2)
        rel[fp8] = ((k_got.double() - k_ref.double()).norm() / k_ref.double().norm()).item()
    print(f"    layer-11 keys rel-L2 vs fp32 oracle: bf16 path {rel[False]:.3e}, fp8-QKV path {rel[True]:.3e}")
    assert rel[False] < 2e-2 and rel[True] < FP8_KEYS_TOL, rel


# ---- the fp8 path's own tolerance table (measured values are printed; bars ~2x the measurement)
FP8_KEYS_TOL = 8e-2
# measured r2 (worst over steps 0-2): keys 4.3e-2; losses <= 8.3e-3 (cls / id / total), <= 1.8e-2 (ssim); generator gradient 9.8e-2
# (ViT-S/8 @ 64) / 3.7e-2 (ViT-B/8 @ 224) -- against 6e-3 / 3e-3 / 7e-3 on the bf16 path
FP8_LOSS_TOL = {"loss": 2e-2, "loss_global_cls": 2e-2, "loss_entire_cls": 2e-2, "loss_global_ssim": 4e-2, "loss_entire_ssim": 4e-2, "loss_global_id_B": 2e-2}
FP8_GRAD_TOL = 2e-1


@pytest.mark.parametrize("name,size", [("dino_vits8", 64), ("dino_vitb8", 224)])
def test_fp8_step_vs_oracle_tolerance_table(name, size):
    from oracle import dino_vit
    from oracle.step import SpliceOracle
    cfg = dict(dino_model_name=name, dino_global_patch_size=size)
    A, B = synth.smooth_image_pair(123, 0, size, size)
    vit_state = synth.vit_params(7, name, img_size=size, w_std=0.05 if size == 64 else 0.03)
    gen_state = synth.generator_params(9, 0.02)
    eng = SpliceEngine(cfg, vit_state, gen_state, (size, size), (size, size), fp8=True)
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
