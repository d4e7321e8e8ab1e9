#!/usr/bin/env python3
"""What the GELU / GELU' epilogues cost on the fc1 / fc2^T shapes (run on the GPU box)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from splice_amd import _lib
L = _lib.lib()
def timed(fn, n=100):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, f = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    f.record(); torch.cuda.synchronize()
    return s.elapsed_time(f) / n * 1e3
for M in (800, 1600, 6400):
    N, K = 3072, 768
    A = torch.randn(M, K, device="cuda").bfloat16(); B = (torch.randn(N, K, device="cuda") * 0.03).bfloat16()
    bias = torch.randn(N, device="cuda") * 0.1
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); pre = torch.empty_like(out)
    g = (torch.randn(M, 768, device="cuda")).bfloat16(); Wt = (torch.randn(N, 768, device="cuda") * 0.03).bfloat16()
    res = {}
    for name, flags, use_pre in (("bias", _lib.EPI_BIAS | _lib.EPI_OUT_BF, False), ("bias+gelu", _lib.EPI_BIAS | _lib.EPI_GELU | _lib.EPI_OUT_BF, False),
                                 ("bias+gelu+pre", _lib.EPI_BIAS | _lib.EPI_GELU | _lib.EPI_OUT_BF, True)):
        e = _lib.GemmEpilogue(); e.bias = bias.data_ptr(); e.out_bf = out.data_ptr(); e.ldbf = N
        if use_pre: e.out_pre = pre.data_ptr(); e.ldp = N
        res[name] = timed(lambda: L.splice_gemm_nt_bf16(flags, _lib.ptr(A), K, _lib.ptr(B), K, M, N, K, C.byref(e), _lib.current_stream()))
    # fc2^T: dh[M][3072] = (g[M][768] @ W[3072][768]^T) * gelu'(pre)
    for name, flags in (("fc2T plain", _lib.EPI_OUT_BF), ("fc2T gelu'", _lib.EPI_GELU_GRAD | _lib.EPI_OUT_BF)):
        e = _lib.GemmEpilogue(); e.out_bf = out.data_ptr(); e.ldbf = N; e.aux = pre.data_ptr(); e.ldaux = N
        res[name] = timed(lambda: L.splice_gemm_nt_bf16(flags, _lib.ptr(g), 768, _lib.ptr(Wt), 768, M, N, 768, C.byref(e), _lib.current_stream()))
    print(f"M={M}: " + " | ".join(f"{k} {v:.1f}us" for k, v in res.items()), flush=True)
