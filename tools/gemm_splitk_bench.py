#!/usr/bin/env python3
"""Split-K study of the plain fp32-output GEMM (run on the GPU box): python tools/gemm_splitk_bench.py"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from splice_amd import _lib
L = _lib.lib()
for name, M, N, K in [("fc2_16", 1600, 768, 3072), ("fc1T8", 800, 768, 3072), ("qkvT8", 800, 768, 2304), ("proj16", 1600, 768, 768), ("fc1_16", 1600, 3072, 768), ("qkv_16", 1600, 2304, 768), ("fc2T8", 800, 3072, 768)]:
    A = torch.randn(M, K, device="cuda").bfloat16(); B = torch.randn(N, K, device="cuda").bfloat16()
    row = []
    for tile in (3, 13, 2, 12, 1, 11):
        for ks in (1, 2, 3, 4, 6):
            if K % (ks * 64): continue
            out = torch.empty(ks, M, N, device="cuda")
            e = _lib.GemmEpilogue(); e.out_f32 = out.data_ptr(); e.ldo = N; e.ksplit = ks; e.slab_stride = M * N
            L.splice_gemm_force_tile(tile)
            f = lambda: L.splice_gemm_nt_bf16(_lib.EPI_OUT_F32, _lib.ptr(A), K, _lib.ptr(B), K, M, N, K, C.byref(e), _lib.current_stream())
            for _ in range(3): f()
            torch.cuda.synchronize()
            s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20): f()
            t.record(); torch.cuda.synchronize()
            row.append(f"t{tile}/ks{ks} {s.elapsed_time(t)/20*1e3:5.1f}")
    L.splice_gemm_force_tile(0)
    print(f"{name:7s} {M}x{N}x{K}: " + " | ".join(row))
