#!/usr/bin/env python3
"""Which kernels does the vendor library pick for the ViT projection shapes?  Run under `rocprofv3 --kernel-trace --stats`:
each shape is called 20 times, preceded by a marker GEMM of a unique odd size so the trace can be cut per shape by order.
Prints nothing useful itself; read the kernel names (Tensile encodes macro tile MT, MFMA MI, workgroup, prefetch depth)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
layers = [("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)]
only = sys.argv[1] if len(sys.argv) > 1 else None
for passes in (2, 16):
    for name, N, K in layers:
        tag = f"{name}_{passes}"
        if only and only != tag:
            continue
        M = passes * 800
        A = torch.randn(M, K, device="cuda").bfloat16()
        W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        for _ in range(20):
            torch.matmul(A, W.t(), out=out)
        torch.cuda.synchronize()
        print(tag, M, N, K, flush=True)
