#!/usr/bin/env python3
"""Workgroup-count quantisation of the 64x64 ring GEMM (run on the GPU box): time vs number of row tiles at N=768."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from splice_amd import _lib
L = _lib.lib()
for K in (3072, 768):
    row = []
    for M in (768, 1024, 1280, 1344, 1408, 1600, 1664, 2048, 2560, 3200):
        N = 768
        A = torch.randn(M, K, device="cuda").bfloat16(); B = torch.randn(N, K, device="cuda").bfloat16()
        out = torch.empty(M, N, device="cuda")
        e = _lib.GemmEpilogue(); e.out_f32 = out.data_ptr(); e.ldo = N
        L.splice_gemm_force_tile(13)
        f = lambda: L.splice_gemm_nt_bf16(_lib.EPI_OUT_F32, _lib.ptr(A), K, _lib.ptr(B), K, M, N, K, C.byref(e), _lib.current_stream())
        for _ in range(5): f()
        torch.cuda.synchronize()
        s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(50): f()
        t.record(); torch.cuda.synchronize()
        row.append(f"M{M}({(M+63)//64*12}wg) {s.elapsed_time(t)/50*1e3:5.1f}")
    L.splice_gemm_force_tile(0)
    print(f"K={K}: " + " | ".join(row))
