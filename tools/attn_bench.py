#!/usr/bin/env python3
"""Micro-benchmark + correctness check of the fused attention kernels on the ViT-B/8 shapes, per kernel
variant (run on the GPU box):  python tools/attn_bench.py [variants, comma separated]

variant = forward queries per wave / 16 (splice_attention_variant: 1 or 2); 0 = the library default.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from splice_amd import _lib

L = _lib.lib()
DEV = "cuda"
variants = [int(t) for t in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0]
std = float(os.environ.get("ATTN_STD", "1.0"))
BWD_VARIANT = int(os.environ.get("ATTN_BWD_VARIANT", "0"))   # splice_attention_bwd_variant: 0 policy, 2 one launch, 3 two launches, 4 two-wave workgroups (pre-scaled q)
FOLD = int(os.environ.get("ATTN_FOLD", "0"))   # 1: q columns pre-multiplied by scale * log2(e) (forward only: the engine's packing of the QKV projection)


def ref_attention(qkv, B, T, Tld, D, H, scale):
    x = qkv.float().reshape(B, Tld, 3, H, D // H)[:, :T].detach().clone().requires_grad_(True)
    q, k, v = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)
    p = ((q @ k.transpose(-1, -2)) * scale).softmax(-1)
    return (p @ v).transpose(1, 2).reshape(B, T, D), x


def relerr(a, b):
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, f = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    f.record()
    torch.cuda.synchronize()
    return s.elapsed_time(f) / n * 1e3


shapes = [(4, 785), (2, 785), (2, 3137)]
if os.environ.get("ATTN_SHAPES"):   # e.g. ATTN_SHAPES=4x785,2x785 (under rocprofv3: one shape per run keeps the per-kernel stats unambiguous)
    shapes = [tuple(int(x) for x in t.split("x")) for t in os.environ["ATTN_SHAPES"].split(",")]
for (B, T) in shapes:
    D, H = 768, 12
    Tld = (T + 31) // 32 * 32
    rows = B * Tld
    scale = (D // H) ** -0.5
    g = torch.Generator(device="cpu").manual_seed(5)
    qkv = (torch.randn(rows, 3 * D, generator=g) * std).to(DEV).bfloat16()
    qkvT = None   # (not read any more)
    dout = torch.randn(B, Tld, D, generator=g)
    dout[:, T:] = 0
    dout = dout.reshape(rows, D).to(DEV).bfloat16()
    doutT = None
    ref = gref = None
    if T < 1000:
        ref, leaf = ref_attention(qkv, B, T, Tld, D, H, scale)
        ref.backward(dout.float().reshape(B, Tld, D)[:, :T])
        gref = leaf.grad
    st = _lib.current_stream()
    fl_f = 4.0 * T * T * 64 * H * B
    qkv_plain = qkv
    if FOLD:
        qkv = qkv.clone()
        qkv[:, :D] = (qkv[:, :D].float() * (scale * 1.4426950408889634)).bfloat16()
    for v in variants:
        L.splice_attention_variant(v)
        out = torch.zeros(rows, D, device=DEV, dtype=torch.bfloat16)
        lse = torch.zeros(B, H, Tld, device=DEV)
        delta = torch.zeros(B, H, Tld, device=DEV)
        dqkv = torch.zeros(rows, 3 * D, device=DEV, dtype=torch.bfloat16)

        kscale = 0.6931471805599453 if FOLD else scale   # kernels fed with pre-scaled q take ln 2 for the scale (scale * q.k == ln2 * q'.k)

        def fwd():
            L.splice_attention_qfold(FOLD)
            _lib.check(L.splice_attention_fwd(_lib.ptr(qkv), _lib.ptr(qkvT), rows, B, T, Tld, D, H, kscale, _lib.ptr(out), _lib.ptr(lse), st))

        def bwd():
            L.splice_attention_qfold(FOLD)
            L.splice_attention_bwd_variant(BWD_VARIANT)
            _lib.check(L.splice_attention_bwd(_lib.ptr(qkv), _lib.ptr(qkvT), rows, B, T, Tld, D, H, kscale, _lib.ptr(out), _lib.ptr(lse),
                                              _lib.ptr(dout), _lib.ptr(doutT), _lib.ptr(delta), _lib.ptr(dqkv), st))

        fwd()
        bwd()
        torch.cuda.synchronize()
        msg = ""
        if ref is not None:
            ef = relerr(out.float().reshape(B, Tld, D)[:, :T], ref)
            gg = dqkv.float().reshape(B, Tld, 3, H, D // H).clone()
            if FOLD:
                gg[:, :, 0] *= scale * 1.4426950408889634   # the kernels return dL/dq'
            eb = [relerr(gg[:, :T, i], gref[:, :, i]) for i in range(3)]
            pad = gg[:, T:, 1:].abs().max().item() if Tld > T else 0.0
            msg = f" err fwd {ef:.1e} dq {eb[0]:.1e} dk {eb[1]:.1e} dv {eb[2]:.1e} pad {pad:g}"
        tf, tb = timed(fwd), timed(bwd)
        print(f"B{B} T{T} variant {v:3d}: fwd {tf:6.1f} us {fl_f / tf / 1e6:5.0f} TF | bwd {tb:6.1f} us {2.5 * fl_f / tb / 1e6:5.0f} TF(alg 2.5x){msg}", flush=True)
    L.splice_attention_variant(0)
