#!/usr/bin/env python3
"""Does the leading dimension of the operands matter (L2 channel camping of power-of-two-ish row strides)?  The auto-dispatched
bf16 NT GEMM on the ViT-B shapes with lda = ldb = K + pad."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from splice_amd import _lib

L = _lib.lib()
pads = [0, 8, 32, 64, 128, 192, 256]
for lname, N, K in (("fc2", 768, 3072), ("fc1", 3072, 768), ("qkv", 2304, 768), ("proj", 768, 768)):
    for M in (800, 1600, 6400, 12800):
        row = []
        for pad in pads:
            ld = K + pad
            A = torch.randn(M, ld, device="cuda").bfloat16()
            B = torch.randn(N, ld, device="cuda").bfloat16()
            out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            e = _lib.GemmEpilogue()
            e.out_bf = out.data_ptr(); e.ldbf = N
            def run(n):
                for _ in range(n):
                    L.splice_gemm_nt_bf16(_lib.EPI_OUT_BF, _lib.ptr(A), ld, _lib.ptr(B), ld, M, N, K, C.byref(e), _lib.current_stream())
            run(3); torch.cuda.synchronize()
            s, f = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); run(30); f.record(); torch.cuda.synchronize()
            row.append(f"+{pad}:{s.elapsed_time(f) / 30 * 1e3:6.1f}")
        print(f"{lname:4s} M={M:6d} N={N:5d} K={K:5d} us by pad: " + " ".join(row), flush=True)
