#!/usr/bin/env python3
"""Micro-benchmark of the bf16 NT GEMM on the ViT shapes, per tile config (run on the GPU box)."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from splice_amd import _lib

L = _lib.lib()
shapes = [("fc2T8", 800, 3072, 768), ("fc1_16", 1600, 3072, 768)] if os.environ.get("GEMM_FC1_ONLY") else [("qkv16", 1600, 2304, 768), ("proj16", 1600, 768, 768), ("fc1_16", 1600, 3072, 768), ("fc2_16", 1600, 768, 3072), ("fc2T8", 800, 3072, 768), ("fc1T8", 800, 768, 3072), ("qkvT8", 800, 768, 2304),
          ("qkv", 3200, 2304, 768), ("proj", 3200, 768, 768), ("fc1", 3200, 3072, 768), ("fc2", 3200, 768, 3072),
          ("fc2T", 1600, 3072, 768), ("fc1T", 1600, 768, 3072), ("projT", 1600, 768, 768), ("qkvT", 1600, 768, 2304),
          ("patch", 3200, 768, 192)]
if os.environ.get("GEMM_SHAPES"):   # e.g. GEMM_SHAPES=fc1T:6400x768x3072,qkvT:6400x768x2304
    shapes = [(t.split(":")[0],) + tuple(int(x) for x in t.split(":")[1].split("x")) for t in os.environ["GEMM_SHAPES"].split(",")]
names = {0: "auto", 1: "128x128", 2: "128x64", 3: "64x64", 11: "128x128r4", 12: "128x64r4", 13: "64x64r4", 23: "64x64r3", 33: "64x64r6x2", 43: "64x96r7", 53: "64x96r4", 63: "64x64r8", 21: "128x128r3", 73: "128x192r3", 83: "64x192r4", 93: "128x96r5"}   # tile + 10 * (4-stage ring)
tiles = [int(t) for t in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 1, 2, 3]
for name, M, N, K in shapes:
    A = torch.randn(M, K, device="cuda").bfloat16()
    B = torch.randn(N, K, device="cuda").bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    e = _lib.GemmEpilogue()
    e.out_bf = out.data_ptr(); e.ldbf = N
    row = []
    ref = A.float() @ B.float().T
    for t in tiles:
        L.splice_gemm_force_tile(t)
        for _ in range(3):
            L.splice_gemm_nt_bf16(_lib.EPI_OUT_BF, _lib.ptr(A), K, _lib.ptr(B), K, M, N, K, C.byref(e), _lib.current_stream())
        torch.cuda.synchronize()
        s, f = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            L.splice_gemm_nt_bf16(_lib.EPI_OUT_BF, _lib.ptr(A), K, _lib.ptr(B), K, M, N, K, C.byref(e), _lib.current_stream())
        f.record(); torch.cuda.synchronize()
        us = s.elapsed_time(f) / 20 * 1e3
        err = ((out.float() - ref).norm() / ref.norm()).item()
        row.append(f"{names.get(t,t)} {us:6.1f}us {2*M*N*K/us/1e6:4.0f}TF{'' if err < 4e-3 else ' ERR %.1e' % err}")
    L.splice_gemm_force_tile(0)
    print(f"{name:6s} {M}x{N}x{K}: " + " | ".join(row))
