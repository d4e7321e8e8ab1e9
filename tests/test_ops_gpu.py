"""GPU parity of every hand-written HIP kernel family, called through the C ABI
(include/splice_hip.h), against a plain fp32 torch / oracle reference of the same op.

Tolerances: bf16 operands (8 mantissa bits) with fp32 accumulation -> inputs are rounded
to bf16 first so the comparison isolates the kernel's own arithmetic; remaining error is
output rounding (bf16 outputs: rel 2^-8) and accumulation order (fp32 outputs: ~1e-5 rel
of the operand magnitude sum).
"""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from splice_amd import _lib, synth

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _st():
    return _lib.current_stream()


def _bf(t):
    return t.to(torch.bfloat16)


def _rand(*shape, seed=0, std=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * std).to(DEV)


def _relerr(a, b):
    return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()


def _gemm(flags, A, B, M, N, K, **kw):
    e = _lib.GemmEpilogue()
    keep = []
    for k, v in kw.items():
        if isinstance(v, torch.Tensor):
            keep.append(v)
            setattr(e, k, v.data_ptr())
        else:
            setattr(e, k, v)
    rc = _lib.lib().splice_gemm_nt_bf16(flags, _lib.ptr(A), A.stride(0), _lib.ptr(B), B.stride(0), M, N, K, C.byref(e), _st())
    _lib.check(rc, "gemm")
    torch.cuda.synchronize()


@pytest.mark.parametrize("M,N,K", [(16, 16, 64), (100, 72, 128), (785, 768, 768), (3140, 2304, 768), (1570, 768, 3072),
                                   (800, 192, 768), (130, 3072, 768)])
def test_gemm_plain_f32(M, N, K):
    A, B = _bf(_rand(M, K, seed=1)), _bf(_rand(N, K, seed=2))
    out = torch.full((M, N), float("nan"), device=DEV)
    _gemm(_lib.EPI_OUT_F32, A, B, M, N, K, out_f32=out, ldo=N)
    ref = A.float() @ B.float().T
    assert torch.isfinite(out).all()
    assert _relerr(out, ref) < 2e-6, _relerr(out, ref)
    # transpose / row-col swap detector: asymmetric operands, compare a few exact entries
    assert torch.allclose(out[M - 1, 0], ref[M - 1, 0], rtol=1e-4, atol=1e-3)
    assert torch.allclose(out[0, N - 1], ref[0, N - 1], rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("force", [1, 11, 2, 12, 22, 3, 13, 23])
def test_gemm_forced_tiles_and_rings(force):
    """Every tile (128x128 / 128x64 / 64x64) x pipeline (2-stage, 4- and 3-stage ring) variant the dispatcher can pick,
    on ragged shapes, through the epilogue with prefetched operands (N % 4 == 0) and its per-fragment fallback (N % 4 != 0)."""
    L = _lib.lib()
    try:
        L.splice_gemm_force_tile(force)
        for M, N, K in [(333, 200, 256), (130, 198, 192), (70, 64, 64)]:
            A, B = _bf(_rand(M, K, seed=11)), _bf(_rand(N, K, seed=12, std=0.05))
            bias, resid = _rand(N, seed=13), _rand(M, N, seed=14)
            out = torch.full((M, N), float("nan"), device=DEV)
            _gemm(_lib.EPI_BIAS | _lib.EPI_RESID | _lib.EPI_OUT_F32, A, B, M, N, K, bias=bias, resid=resid, ldr=N, resid_mod=0, out_f32=out, ldo=N)
            ref = A.float() @ B.float().T + bias + resid
            assert torch.isfinite(out).all(), (force, M, N, K)
            assert _relerr(out, ref) < 2e-6, (force, M, N, K, _relerr(out, ref))
    finally:
        L.splice_gemm_force_tile(0)


@pytest.mark.parametrize("M,N,K", [(700, 512, 256), (1000, 768, 128), (6400, 2304, 768)])
def test_gemm_8phase_tile_is_bit_identical(M, N, K):
    """The persistent 256 x 256 8-phase tile (gemm8p.h; forced with tile code 5, and selected by the dispatcher itself at the batched
    shape 6400 x 2304 x 768 = QKV at four pairs per GPU) gives the SAME bf16 bits as the 128 x 128 tile for both of its epilogues:
    bias -> bf16 (QKV) and bias -> [bf16 pre-activation of the rows >= pre_row_lo] -> GELU -> bf16 (fc1); ragged M, several tiles per
    workgroup (the stream of K tiles runs on across output tiles), rows outside the matrix never stored."""
    L = _lib.lib()
    A, B = _bf(_rand(M, K, seed=31)), _bf(_rand(N, K, seed=32, std=0.05))
    bias = _rand(N, seed=33)
    lo = M // 3
    outs = {}
    try:
        for force in (1, 5, 0):
            L.splice_gemm_force_tile(force)
            ob = torch.full((M + 64, N), -7.0, device=DEV, dtype=torch.bfloat16)       # guard rows behind the matrix
            _gemm(_lib.EPI_BIAS | _lib.EPI_OUT_BF, A, B, M, N, K, bias=bias, out_bf=ob, ldbf=N)
            og = torch.full((M + 64, N), -7.0, device=DEV, dtype=torch.bfloat16)
            opre = torch.full((M + 64, N), -7.0, device=DEV, dtype=torch.bfloat16)
            _gemm(_lib.EPI_BIAS | _lib.EPI_GELU | _lib.EPI_OUT_BF, A, B, M, N, K, bias=bias, out_bf=og, ldbf=N, out_pre=opre, ldp=N, pre_row_lo=lo)
            of = torch.full((M + 64, N), -7.0, device=DEV)
            resid = _rand(M, N, seed=34)
            _gemm(_lib.EPI_BIAS | _lib.EPI_RESID | _lib.EPI_OUT_F32, A, B, M, N, K, bias=bias, resid=resid, ldr=N, resid_mod=0, out_f32=of, ldo=N)
            outs[force] = (ob, og, opre, of)
    finally:
        L.splice_gemm_force_tile(0)
    ref = A.float() @ B.float().T + bias
    assert _relerr(outs[1][0][:M].float(), ref) < 3e-3
    for force in (5, 0):
        for a, b, name in zip(outs[force], outs[1], ("qkv", "gelu", "pre", "bias + residual -> fp32 (proj / fc2)")):
            assert torch.equal(a[:M], b[:M]), (force, name, (a[:M].float() - b[:M].float()).abs().max().item())
            assert (a[M:] == -7.0).all(), (force, name, "rows behind the matrix were written")
    assert (outs[5][2][:lo] == -7.0).all() and (outs[5][2][lo:M] != -7.0).any()      # pre-activation only for the gradient-carrying rows


@pytest.mark.parametrize("K", [768, 3072])
def test_gemm_one_wave_tiles(K):
    """2 x 785 tokens, N = 768 with the bias + residual epilogue (proj / fc2 forward): for K = 768 the dispatcher takes one
    wave of 64x96 tiles (200 workgroups) instead of 300 64x64 ones."""
    M, N = 1570, 768
    A, B = _bf(_rand(M, K, seed=21)), _bf(_rand(N, K, seed=22, std=0.05))
    bias, resid = _rand(N, seed=23), _rand(M, N, seed=24)
    out = torch.full((M, N), float("nan"), device=DEV)
    _gemm(_lib.EPI_BIAS | _lib.EPI_RESID | _lib.EPI_OUT_F32, A, B, M, N, K, bias=bias, resid=resid, ldr=N, resid_mod=0, out_f32=out, ldo=N)
    ref = A.float() @ B.float().T + bias + resid
    assert torch.isfinite(out).all()
    assert _relerr(out, ref) < 2e-6, _relerr(out, ref)
    assert torch.allclose(out[M - 1, 0], ref[M - 1, 0], rtol=1e-4, atol=1e-3) and torch.allclose(out[0, N - 1], ref[0, N - 1], rtol=1e-4, atol=1e-3)


def test_gemm_epilogues():
    M, N, K = 785, 768, 768
    A, B = _bf(_rand(M, K, seed=3)), _bf(_rand(N, K, seed=4, std=0.05))
    bias = _rand(N, seed=5)
    resid = _rand(M, N, seed=6)
    base = A.float() @ B.float().T
    # bias + residual -> fp32
    out = torch.empty(M, N, device=DEV)
    _gemm(_lib.EPI_BIAS | _lib.EPI_RESID | _lib.EPI_OUT_F32, A, B, M, N, K, bias=bias, resid=resid, ldr=N, resid_mod=0,
          out_f32=out, ldo=N)
    assert _relerr(out, base + bias + resid) < 2e-6
    # residual with row modulo (pos-embed style)
    pos = _rand(200, N, seed=7)
    _gemm(_lib.EPI_BIAS | _lib.EPI_RESID | _lib.EPI_OUT_F32, A, B, M, N, K, bias=bias, resid=pos, ldr=N, resid_mod=200,
          out_f32=out, ldo=N)
    ref = base + bias + pos[torch.arange(M, device=DEV) % 200]
    assert _relerr(out, ref) < 2e-6
    # bias -> bf16 + transposed bf16 + fp32 column window
    ldt = 788
    ob = torch.zeros(M, N, device=DEV, dtype=torch.bfloat16)
    obt = torch.zeros(N, ldt, device=DEV, dtype=torch.bfloat16)
    oc = torch.zeros(M, 256, device=DEV)
    _gemm(_lib.EPI_BIAS | _lib.EPI_OUT_BF | _lib.EPI_OUT_T | _lib.EPI_COLS_F32, A, B, M, N, K, bias=bias, out_bf=ob,
          ldbf=N, out_bf_t=obt, ldt=ldt, out_f32_cols=oc, ld_cols=256, col_lo=256, col_hi=512)
    ref = base + bias
    assert _relerr(ob.float(), ref) < 3e-3
    assert torch.equal(obt[:, :M].T.contiguous(), ob)
    assert _relerr(oc, ref[:, 256:512]) < 2e-6
    # GELU (+ saved pre-activation) and the GELU-grad epilogue
    og = torch.zeros(M, N, device=DEV, dtype=torch.bfloat16)
    opre = torch.zeros(M, N, device=DEV, dtype=torch.bfloat16)
    _gemm(_lib.EPI_BIAS | _lib.EPI_GELU | _lib.EPI_OUT_BF, A, B, M, N, K, bias=bias, out_bf=og, ldbf=N, out_pre=opre, ldp=N)
    assert _relerr(opre.float(), ref) < 3e-3
    assert _relerr(og.float(), torch.nn.functional.gelu(ref)) < 4e-3
    od = torch.zeros(M, N, device=DEV, dtype=torch.bfloat16)
    _gemm(_lib.EPI_GELU_GRAD | _lib.EPI_OUT_BF, A, B, M, N, K, aux=opre, ldaux=N, out_bf=od, ldbf=N)
    x = opre.float().requires_grad_(True)
    torch.nn.functional.gelu(x).backward(base)
    assert _relerr(od.float(), x.grad) < 4e-3


def test_layernorm():
    rows, D = 1571, 768
    x = _rand(rows, D, seed=8, std=3.0) + 0.5
    gamma, beta = 1 + 0.1 * _rand(D, seed=9), 0.1 * _rand(D, seed=10)
    y = torch.empty(rows, D, device=DEV, dtype=torch.bfloat16)
    mean, rstd = torch.empty(rows, device=DEV), torch.empty(rows, device=DEV)
    L = _lib.lib()
    _lib.check(L.splice_layernorm_fwd(_lib.ptr(x), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(y), _lib.ptr(mean),
                                      _lib.ptr(rstd), rows, D, 1e-6, _st()))
    xr = x.clone().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (D,), gamma, beta, 1e-6)
    assert _relerr(y.float(), ref) < 3e-3
    assert _relerr(mean, x.mean(1)) < 1e-5
    dy = _rand(rows, D, seed=11)
    g_in = _rand(rows, D, seed=12)
    g_out = torch.empty(rows, D, device=DEV)
    g_bf = torch.empty(rows, D, device=DEV, dtype=torch.bfloat16)
    _lib.check(L.splice_layernorm_bwd(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(gamma), _lib.ptr(mean), _lib.ptr(rstd),
                                      _lib.ptr(g_in), _lib.ptr(g_out), _lib.ptr(g_bf), rows, D, _st()))
    ref.backward(dy)
    assert _relerr(g_out, xr.grad + g_in) < 1e-5
    assert _relerr(g_bf.float(), xr.grad + g_in) < 3e-3
    # D = 384 (ViT-S)
    x2 = _rand(70, 384, seed=13)
    y2 = torch.empty(70, 384, device=DEV, dtype=torch.bfloat16)
    _lib.check(L.splice_layernorm_fwd(_lib.ptr(x2), _lib.ptr(gamma[:384].contiguous()), _lib.ptr(beta[:384].contiguous()),
                                      _lib.ptr(y2), None, None, 70, 384, 1e-6, _st()))
    assert _relerr(y2.float(), torch.nn.functional.layer_norm(x2, (384,), gamma[:384], beta[:384], 1e-6)) < 3e-3


def _attn_ref(qkv, B, T, Tld, D, H, scale):
    """fp32 reference on the bf16-rounded qkv; returns out [B,T,D] and the leaf for autograd."""
    x = qkv.float().reshape(B, Tld, 3, H, D // H)[:, :T].detach().clone().requires_grad_(True)
    q, k, v = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)
    p = ((q @ k.transpose(-1, -2)) * scale).softmax(-1)
    o = (p @ v).transpose(1, 2).reshape(B, T, D)
    return o, x, p


def test_attention_forward_score_jump_takes_the_exact_walk():
    """The 32x32x16 forward (attn_x32.h) forms a tile's probabilities against the reference point in force and looks at the row
    sums AFTER the tile (lazy deferred maximum).  A score that exceeds everything seen in earlier tiles by more than 2^97 inside
    one tile overflows fp32 before that look; the kernel must notice (non-finite row sum) and recompute those queries with the
    exact walk.  Here: one key, far into the sequence, whose score against some queries is ~18000 log2 units above the rest."""
    B, T, D, H = 2, 785, 768, 12
    Tld = (T + 31) // 32 * 32
    rows = B * Tld
    scale = (D // H) ** -0.5
    x = _rand(rows, 3 * D, seed=77, std=1.0)
    for (b, q, h, key) in [(0, 5, 3, 300), (1, 700, 11, 784), (1, 64, 0, 65)]:
        x[b * Tld + q, h * 64:(h + 1) * 64] = 40.0
        x[b * Tld + key, D + h * 64:D + (h + 1) * 64] = 40.0
    qkv = _bf(x)
    L = _lib.lib()
    ref, _, _ = _attn_ref(qkv, B, T, Tld, D, H, scale)
    for variant in (0, 42):
        L.splice_attention_variant(variant)
        out = torch.zeros(rows, D, device=DEV, dtype=torch.bfloat16)
        lse = torch.zeros(B, H, Tld, device=DEV)
        _lib.check(L.splice_attention_fwd(_lib.ptr(qkv), None, 0, B, T, Tld, D, H, scale, _lib.ptr(out), _lib.ptr(lse), _st()))
        torch.cuda.synchronize()
        L.splice_attention_variant(0)
        got = out.float().reshape(B, Tld, D)[:, :T]
        assert torch.isfinite(got).all() and torch.isfinite(lse[:, :, :T]).all(), variant
        assert _relerr(got, ref.detach()) < 6e-3, (variant, _relerr(got, ref.detach()))
        # the spiked queries attend to their key alone
        v = qkv.float().reshape(B, Tld, 3, H, 64)
        assert torch.allclose(got[0, 5].reshape(H, 64)[3], v[0, 300, 2, 3], atol=2e-2)


# (1, 3137, ...) = the 448x448 sequence length of BASELINE configs[3]: 50 key tiles per query block and the
# two-launch backward (attn_bwd_q_kernel + attn_bwd_kv_kernel), which the T <= 785 cases never reach
@pytest.mark.parametrize("B,T,D,H,std", [(1, 17, 384, 6, 1.0), (2, 197, 768, 12, 1.0), (2, 785, 768, 12, 0.6), (1, 785, 768, 12, 2.5),
                                         (1, 3137, 768, 12, 0.6), (2, 1601, 768, 12, 1.0)])
def test_attention_fwd_bwd(B, T, D, H, std):
    Tld = (T + 31) // 32 * 32
    rows = B * Tld
    scale = (D // H) ** -0.5
    qkv = _bf(_rand(rows, 3 * D, seed=20, std=std))
    qkvT = None   # the transposed copies are not read any more (transposing LDS reads): the ABI allows NULL / 0 for them
    out = torch.zeros(rows, D, device=DEV, dtype=torch.bfloat16)
    lse = torch.zeros(B, H, Tld, device=DEV)
    L = _lib.lib()
    _lib.check(L.splice_attention_fwd(_lib.ptr(qkv), _lib.ptr(qkvT), 0, B, T, Tld, D, H, scale, _lib.ptr(out),
                                      _lib.ptr(lse), _st()))
    torch.cuda.synchronize()
    ref, leaf, p = _attn_ref(qkv, B, T, Tld, D, H, scale)
    got = out.float().reshape(B, Tld, D)[:, :T]
    assert torch.isfinite(out.float()).all()
    assert _relerr(got, ref) < 6e-3, _relerr(got, ref)
    # launch forms must agree BIT FOR BIT -- a pass's attention output may not depend on how many passes share the launch.  The bf16 forward is the
    # 32x32x16 kernel (attn_x32.h); variants 41 / 42 / 48 = 4 / 2 / 8 waves per workgroup (the 16x16x32 forms of rounds 1-4 left the library in round 6)
    def run(variant):
        L.splice_attention_variant(variant)
        out_v, lse_v = torch.zeros_like(out), torch.zeros_like(lse)
        _lib.check(L.splice_attention_fwd(_lib.ptr(qkv), _lib.ptr(qkvT), rows, B, T, Tld, D, H, scale, _lib.ptr(out_v), _lib.ptr(lse_v), _st()))
        torch.cuda.synchronize()
        L.splice_attention_variant(0)
        return out_v, lse_v
    for variant in (41, 42, 48):
        out_v, lse_v = run(variant)
        assert torch.equal(out_v, out) and torch.equal(lse_v, lse), variant
    # probabilities API
    probs = torch.empty(B, H, T, T, device=DEV)
    _lib.check(L.splice_attention_probs(_lib.ptr(qkv), B, T, Tld, D, H, scale, _lib.ptr(lse), _lib.ptr(probs), _st()))
    assert _relerr(probs, p) < 2e-3
    # backward; padded query rows carry zero upstream gradient (engine invariant)
    dout = _rand(B, Tld, D, seed=21)
    dout[:, T:] = 0
    dout = _bf(dout.reshape(rows, D))
    doutT = None
    delta = torch.zeros(B, H, Tld, device=DEV)
    dqkv = torch.zeros(rows, 3 * D, device=DEV, dtype=torch.bfloat16)
    _lib.check(L.splice_attention_bwd(_lib.ptr(qkv), _lib.ptr(qkvT), 0, B, T, Tld, D, H, scale, _lib.ptr(out),
                                      _lib.ptr(lse), _lib.ptr(dout), _lib.ptr(doutT), _lib.ptr(delta), _lib.ptr(dqkv), _st()))
    torch.cuda.synchronize()
    ref.backward(dout.float().reshape(B, Tld, D)[:, :T])
    gref = leaf.grad  # [B,T,3,H,d]
    ggot = dqkv.float().reshape(B, Tld, 3, H, D // H)
    assert torch.isfinite(ggot).all()
    for i, name in enumerate("qkv"):
        e = _relerr(ggot[:, :T, i], gref[:, :, i])
        assert e < 2e-2, (name, e)
    assert ggot[:, T:, 1:].abs().max().item() == 0.0  # padded keys/values get exactly zero gradient



# What the ENGINE runs (VERDICT r5 "What's weak" #1): q pre-scaled by scale * log2(e) (the ViT engine packs the q rows of the QKV projection that
# way), forward attn_fwd_x32_kernel<., FOLD = true>, backward attn_bwd_x32_kernel in ONE launch (<= 1400 workgroups), in TWO launches
# (attn_bwd_q_x32_kernel + attn_bwd_kv_x32_kernel: 9+ passes at T = 785, 3+ passes at T = 3137) or as two-wave workgroups.  (9, 785) takes the
# two-launch form by the POLICY; every case also forces all three forms and compares bits.
@pytest.mark.parametrize("B,T,D,H,std", [(1, 17, 384, 6, 1.0), (2, 197, 768, 12, 1.0), (2, 785, 768, 12, 0.6), (1, 785, 768, 12, 2.5),
                                         (9, 785, 768, 12, 1.0), (2, 3137, 768, 12, 0.6), (2, 1601, 768, 12, 1.0)])
def test_attention_fwd_bwd_prescaled_q_engine_forms(B, T, D, H, std):
    Tld = (T + 31) // 32 * 32
    rows = B * Tld
    scale = (D // H) ** -0.5
    c = scale * 1.4426950408889634
    x = _rand(rows, 3 * D, seed=23, std=std)
    x[:, :D] *= c                       # the stored q' = q * scale * log2(e), rounded to bf16 ONCE (vit_engine.hip pack_qkv)
    qkv = _bf(x)
    qkv_eff = qkv.float().clone()
    qkv_eff[:, :D] /= c                 # the q the fp32 reference sees
    L = _lib.lib()
    ln2 = 0.6931471805599453
    L.splice_attention_qfold(1)
    try:
        def fwd(variant):
            L.splice_attention_variant(variant)
            o, l = torch.zeros(rows, D, device=DEV, dtype=torch.bfloat16), torch.zeros(B, H, Tld, device=DEV)
            _lib.check(L.splice_attention_fwd(_lib.ptr(qkv), None, rows, B, T, Tld, D, H, ln2, _lib.ptr(o), _lib.ptr(l), _st()))
            torch.cuda.synchronize()
            L.splice_attention_variant(0)
            return o, l
        out, lse = fwd(0)
        ref, leaf, _ = _attn_ref(qkv_eff, B, T, Tld, D, H, scale)
        got = out.float().reshape(B, Tld, D)[:, :T]
        assert torch.isfinite(out.float()).all()
        assert _relerr(got, ref) < 6e-3, _relerr(got, ref)
        for variant in (41, 42, 48):
            o_v, l_v = fwd(variant)
            assert torch.equal(o_v, out) and torch.equal(l_v, lse), variant
        dout = _rand(B, Tld, D, seed=24)
        dout[:, T:] = 0
        dout = _bf(dout.reshape(rows, D))

        def bwd(variant):
            L.splice_attention_bwd_variant(variant)
            delta = torch.zeros(B, H, Tld, device=DEV)
            dqkv = torch.zeros(rows, 3 * D, device=DEV, dtype=torch.bfloat16)
            _lib.check(L.splice_attention_bwd(_lib.ptr(qkv), None, rows, B, T, Tld, D, H, ln2, _lib.ptr(out), _lib.ptr(lse), _lib.ptr(dout), None,
                                              _lib.ptr(delta), _lib.ptr(dqkv), _st()))
            torch.cuda.synchronize()
            L.splice_attention_bwd_variant(0)
            return dqkv
        dqkv = bwd(0)
        ref.backward(dout.float().reshape(B, Tld, D)[:, :T])
        gref = leaf.grad   # [B,T,3,H,d] with respect to the un-scaled q
        ggot = dqkv.float().reshape(B, Tld, 3, H, D // H).clone()
        ggot[:, :, 0] *= c   # the kernels return dL/dq'; q = q'/c
        assert torch.isfinite(ggot).all()
        for i, name in enumerate("qkv"):
            e = _relerr(ggot[:, :T, i], gref[:, :, i])
            assert e < 2e-2, (name, e)
        assert ggot[:, T:, 1:].abs().max().item() == 0.0
        for variant in (2, 3, 4):   # one launch / two launches / two-wave workgroups: same bodies, same bits
            assert torch.equal(bwd(variant), dqkv), variant
    finally:
        L.splice_attention_qfold(0)
        L.splice_attention_variant(0)
        L.splice_attention_bwd_variant(0)

def _selfsim(K, T, D, eps=1e-8):
    L = _lib.lib()
    ws = torch.empty(L.splice_keys_selfsim_ws_bytes(T, D), device=DEV, dtype=torch.uint8)
    S = torch.empty(T, T, device=DEV)
    _lib.check(L.splice_keys_selfsim_fwd(_lib.ptr(K), K.stride(0), T, D, eps, _lib.ptr(S), _lib.ptr(ws), _st()))
    return S, ws


def test_selfsim_golden(golden_dir):
    """attn_cosine_sim against the vectors recorded from the reference function."""
    g = np.load(os.path.join(golden_dir, "extractor.npz"))
    for T, D, key in ((197, 64, "cos_T197_D64"),):
        x = torch.from_numpy(synth.normal(11, f"cos/{T}", (1, 1, T, D)))[0, 0].to(DEV)
        S, _ = _selfsim(x, T, D)
        ref = torch.from_numpy(g[key])[0].to(DEV)
        # bf16 operands: |err| <= ~2^-8 on values in [-1,1]
        assert (S - ref).abs().max().item() < 8e-3
        assert (S.diagonal() - 1).abs().max().item() < 1e-5


@pytest.mark.parametrize("T,D", [(65, 384), (197, 768), (785, 768)])
def test_selfsim_fwd_bwd(T, D):
    from oracle.extractor import attn_cosine_sim
    K = _rand(T, D, seed=30) * (1 + _rand(T, 1, seed=31).abs())
    Kb = _bf(K).float()  # the kernel rounds K to bf16; compare on the rounded operand
    S, ws = _selfsim(K, T, D)
    leaf = Kb.clone().requires_grad_(True)
    ref = attn_cosine_sim(leaf[None, None])[0]
    assert (S - ref).abs().max().item() < 2e-5
    dS = _rand(T, T, seed=32)  # deliberately NOT symmetric
    dK = torch.full((T, D + 8), float("nan"), device=DEV)
    L = _lib.lib()
    _lib.check(L.splice_keys_selfsim_bwd(_lib.ptr(dS), _lib.ptr(S), T, D, 1e-8, _lib.ptr(dK), D + 8, 0, _lib.ptr(ws), _st()))
    ref.backward(dS)
    e = _relerr(dK[:, :D], leaf.grad)
    assert e < 1e-2, e
    # accumulate mode
    _lib.check(L.splice_keys_selfsim_bwd(_lib.ptr(dS), _lib.ptr(S), T, D, 1e-8, _lib.ptr(dK), D + 8, 1, _lib.ptr(ws), _st()))
    assert _relerr(dK[:, :D], 2 * leaf.grad) < 1e-2


def test_mse():
    a, b = _rand(785, 800, seed=40), _rand(785, 790, seed=41)
    loss = torch.zeros(1, device=DEV)
    grad = torch.zeros(785, 785, device=DEV)
    _lib.check(_lib.lib().splice_mse(_lib.ptr(a), 800, _lib.ptr(b), 790, 785, 785, 10.0, _lib.ptr(loss), _lib.ptr(grad), 785, _st()))
    d = a[:, :785] - b[:, :785]
    assert abs(loss.item() - 10.0 * (d * d).mean().item()) < 1e-4 * loss.item()
    assert _relerr(grad, 10.0 * 2 * d / d.numel()) < 1e-6


@pytest.mark.parametrize("p,H,W", [(8, 32, 48), (16, 64, 64), (8, 224, 224)])
def test_patchify_roundtrip(p, H, W):
    B = 2
    T = 1 + (H // p) * (W // p)
    Tld = (T + 31) // 32 * 32
    img = torch.rand(B, 3, H, W, device=DEV)
    pat = torch.full((B * Tld, 3 * p * p), 7.0, device=DEV, dtype=torch.bfloat16)
    L = _lib.lib()
    _lib.check(L.splice_patchify(_lib.ptr(img), _lib.ptr(pat), B, H, W, p, Tld, 1, _st()))
    mean = torch.tensor([0.485, 0.456, 0.406], device=DEV).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225], device=DEV).view(1, 3, 1, 1)
    ref = torch.nn.functional.unfold((img - mean) / std, p, stride=p).transpose(1, 2)  # [B, T-1, 3pp]
    got = pat.float().reshape(B, Tld, -1)
    assert _relerr(got[:, 1:T], ref) < 3e-3
    assert got[:, 0].abs().max() == 0 and got[:, T:].abs().max() == 0
    dp = torch.randn(B * Tld, 3 * p * p, device=DEV)
    dimg = torch.empty(B, 3, H, W, device=DEV)
    _lib.check(L.splice_unpatchify(_lib.ptr(dp), _lib.ptr(dimg), B, H, W, p, Tld, 1, _st()))
    refd = torch.nn.functional.fold(dp.reshape(B, Tld, -1)[:, 1:T].transpose(1, 2).contiguous(), (H, W), p, stride=p) / std
    assert _relerr(dimg, refd) < 1e-6


def test_augment_structure_hip_vs_torch():
    """splice_augment_structure (flip / ColorJitter in every op order / GaussianBlur) against the torch restatement of the
    same torchvision-0.10 arithmetic (splice_amd/augment.py, itself pinned against PIL / colorsys on the CPU)."""
    from splice_amd import augment
    img = torch.from_numpy(synth.smooth_image_pair(6, 0, 57, 83)[0]).to(DEV)
    torch.manual_seed(4)
    n_jit = n_blur = n_flip = 0
    for _ in range(60):
        flip, jitter, sigma = augment.draw_structure_params()
        ref = augment.apply_structure_torch(img, flip, jitter, sigma)
        got = augment.apply_structure_hip(img, flip, jitter, sigma)
        assert got.shape == ref.shape
        err = (got - ref).abs().max().item()
        assert err < 2e-5, (flip, jitter, sigma, err)   # hue: two divisions + a 6-way select per pixel
        n_jit += jitter is not None; n_blur += sigma is not None; n_flip += flip
    assert n_jit > 15 and n_blur > 3 and n_flip > 15
    # every position of the contrast op in the chain (the op list is cut there)
    for order in ([1, 0, 2, 3], [0, 1, 3, 2], [3, 2, 1, 0], [2, 3, 0, 1]):
        jitter = (order, (1.3, 0.7, 1.15, -0.08))
        err = (augment.apply_structure_hip(img, True, jitter, 1.1) - augment.apply_structure_torch(img, True, jitter, 1.1)).abs().max().item()
        assert err < 2e-5, (order, err)


def test_splitk_virtual_equals_real_bitwise():
    """Split-K dgrads: the few-row shapes write ks slabs that the consumer adds in order; from 2401 rows on ONE workgroup walks
    the whole K, keeps the ks slab sums apart and adds them in the same order.  Row r of a big call must equal, bit for bit,
    the in-order slab sum of the same row computed in a small call (this is what keeps a pair's gradient independent of the
    number of pairs per step)."""
    L = _lib.lib()
    N, K, ks = 768, 3072, 3
    A = _bf(_rand(3200, K, seed=40))
    B = _bf(_rand(N, K, seed=41, std=0.05))
    assert L.splice_gemm_splitk_slabs(800, ks) == ks and L.splice_gemm_splitk_slabs(3200, ks) == 1
    big = torch.zeros(3200, N, device=DEV)
    e = _lib.GemmEpilogue(); e.out_f32 = big.data_ptr(); e.ldo = N; e.ksplit = ks; e.slab_stride = 3200 * N
    _lib.check(L.splice_gemm_nt_bf16(_lib.EPI_OUT_F32, _lib.ptr(A), K, _lib.ptr(B), K, 3200, N, K, C.byref(e), _st()))
    for r0 in (0, 1600, 2400):
        Ar = A[r0:r0 + 800].contiguous()
        slabs = torch.zeros(ks, 800, N, device=DEV)
        e2 = _lib.GemmEpilogue(); e2.out_f32 = slabs.data_ptr(); e2.ldo = N; e2.ksplit = ks; e2.slab_stride = 800 * N
        _lib.check(L.splice_gemm_nt_bf16(_lib.EPI_OUT_F32, _lib.ptr(Ar), K, _lib.ptr(B), K, 800, N, K, C.byref(e2), _st()))
        torch.cuda.synchronize()
        ref = (slabs[0] + slabs[1]) + slabs[2]
        assert torch.equal(big[r0:r0 + 800], ref), (r0, (big[r0:r0 + 800] - ref).abs().max().item())
    assert _relerr(big, A.float() @ B.float().T) < 1e-5


@pytest.mark.parametrize("case,tol", [("up128", 2e-5), ("down900", 1e-4), ("cap64x150", 2e-5)])
def test_resize_kernels_against_independent_numpy_restatement(case, tol):
    """``resize_bilinear_fwd`` (the Resize of util/losses.py:20 inside the fused step) against oracle/resize_np.py's fixtures (torchvision 0.10
    tensor Resize restated in numpy / float64, tests/golden/resize_np.npz): 128 -> 224, the reference's default 900 -> 224 down-scale without
    antialias, and the 64 x 150 -> 204 x 480 max_size cap.  The adjoint kernel is pinned by the size-independent property
    <R x, y> == <x, R^T y>."""
    from oracle.make_resize_golden import CASES, case_input
    shape, size = CASES[case]
    want = torch.from_numpy(np.load(os.path.join(os.path.dirname(__file__), "golden", "resize_np.npz"))[case]).to(DEV)
    x = torch.from_numpy(case_input(case)).to(DEV).contiguous()
    c, h, w = x.shape
    oh, ow = want.shape[-2:]
    y = torch.empty(c, oh, ow, device=DEV)
    L = _lib.lib()
    _lib.check(L.splice_resize_bilinear_fwd(_lib.ptr(x), _lib.ptr(y), c, h, w, oh, ow, _st()))
    torch.cuda.synchronize()
    assert (y - want).abs().max().item() < tol, (y - want).abs().max().item()
    dy = _rand(c, oh, ow, seed=5)
    dx = torch.zeros(c, h, w, device=DEV)
    _lib.check(L.splice_resize_bilinear_bwd(_lib.ptr(dy), _lib.ptr(dx), c, h, w, oh, ow, _st()))
    torch.cuda.synchronize()
    lhs, rhs = (y.double() * dy.double()).sum().item(), (x.double() * dx.double()).sum().item()
    assert abs(lhs - rhs) < 1e-5 * max(1.0, abs(lhs)), (lhs, rhs)
