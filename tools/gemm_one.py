#!/usr/bin/env python3
"""Run one GEMM shape repeatedly (for rocprofv3 --pmc): python tools/gemm_one.py M N K tile reps"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from splice_amd import _lib
M, N, K, tile, reps = [int(x) for x in sys.argv[1:6]]
L = _lib.lib()
A = torch.randn(M, K, device="cuda").bfloat16(); B = torch.randn(N, K, device="cuda").bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
e = _lib.GemmEpilogue(); e.out_bf = out.data_ptr(); e.ldbf = N
L.splice_gemm_force_tile(tile)
for _ in range(reps):
    L.splice_gemm_nt_bf16(_lib.EPI_OUT_BF, _lib.ptr(A), K, _lib.ptr(B), K, M, N, K, C.byref(e), _lib.current_stream())
torch.cuda.synchronize()
