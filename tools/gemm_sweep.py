#!/usr/bin/env python3
"""Tile / ring sweep of the bf16 NT GEMM over the row counts of P-pair batches (run on the GPU box):
M = rows of one launch (forward: 2P passes, backward: P passes of 800 rows), per (N, K) of the ViT-B layers."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from splice_amd import _lib

L = _lib.lib()
names = {0: "auto", 1: "128x128", 2: "128x64", 3: "64x64", 11: "128x128r4", 12: "128x64r4", 13: "64x64r4", 22: "128x64r3", 23: "64x64r3"}
tiles = [0, 1, 2, 3, 11, 12, 13, 22, 23]
Ms = [int(x) for x in os.environ.get("SWEEP_M", "800,1600,3200,6400,12800").split(",")]
for lname, N, K in (("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072), ("qkvT", 768, 2304)):
    for M in Ms:
        A = torch.randn(M, K, device="cuda").bfloat16()
        B = torch.randn(N, K, device="cuda").bfloat16()
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        e = _lib.GemmEpilogue()
        e.out_bf = out.data_ptr(); e.ldbf = N
        row = []
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
            row.append((2 * M * N * K / us / 1e6, names[t], us))
        L.splice_gemm_force_tile(0)
        best = max(row[1:])
        print(f"{lname:5s} M={M:6d} N={N:5d} K={K:5d}: auto {row[0][0]:4.0f}TF {row[0][2]:6.1f}us | best {best[1]:10s} {best[0]:4.0f}TF {best[2]:6.1f}us | " +
              " ".join(f"{n}:{tf:.0f}" for tf, n, _ in row[1:]), flush=True)
