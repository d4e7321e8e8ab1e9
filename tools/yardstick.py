#!/usr/bin/env python3
"""External yardstick (VERDICT r5 #1a): the vendor library on the SAME shapes as our own kernels, same box, same process.

  * `torch.matmul` (hipBLASLt / rocBLAS behind ATen) against `splice_gemm_nt_bf16` on the eight ViT-B projection shapes
    at M = 1570 / 3140 / 6280 / 12560 token rows (our engine pads a pass to Tld = 800 rows: it is timed at M = 1600 ...
    and the library at BOTH the padded and the unpadded row count);
  * `F.scaled_dot_product_attention` (whatever backend ATen picks on ROCm: flash / efficient / math, each forced in turn
    where it is available) against `splice_attention_fwd / bwd` at (B, T) = (4, 785), (32, 785), (4, 3137).

Tools only: nothing in the product path imports torch.matmul or SDPA.  GEMM timing = 20 calls captured into one hipGraph and
replayed (both columns alike): the first version of this table timed eager calls and read 17.7 us for EVERY small library
GEMM -- torch's host path, not the kernel.  Attention timing = HIP events around back-to-back eager calls (the kernels are
long enough for the host to stay ahead).
Output: a table on stdout and `gpurun_out/yardstick.json`.
"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from splice_amd import _lib

L = _lib.lib()
DEV = "cuda"


def timed(fn, n=50, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        s, f = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            fn()
        f.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(f) / n * 1e3)
    return best


def timed_graph(fn, n=20):
    """Kernel-bound time of `fn`: n calls captured into ONE hipGraph and replayed -- torch's per-call host path (~18 us for
    torch.matmul here) is out of the measurement, what remains is kernel time + the dependent-launch gap inside a graph."""
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3):
            fn()
        st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(n):
                fn()
        for _ in range(3):
            g.replay()
        st.synchronize()
        best = 1e30
        for _ in range(5):
            s, f = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(st)
            g.replay()
            f.record(st)
            st.synchronize()
            best = min(best, s.elapsed_time(f) / n * 1e3)
    return best


def gemm_table():
    rows = []
    # (name, N, K): forward projections and their data gradients (NT form: out[M][N] = A[M][K] . W[N][K]^T)
    layers = [("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072),
              ("fc2T", 3072, 768), ("fc1T", 768, 3072), ("projT", 768, 768), ("qkvT", 768, 2304)]
    for passes in (2, 4, 8, 16):
        for name, N, K in layers:
            M_pad, M_raw = passes * 800, passes * 785
            W = (torch.randn(N, K, device=DEV) * 0.05).bfloat16()
            rec = dict(kind="gemm", layer=name, passes=passes, N=N, K=K)
            for tag, M in (("pad", M_pad), ("raw", M_raw)):
                A = torch.randn(M, K, device=DEV).bfloat16()
                out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
                Wt = W.t()
                us_lib = timed_graph(lambda: torch.matmul(A, Wt, out=out))
                # F.linear takes the weight as [N][K] directly (no transposed view): lets ATen pick its own layout path
                us_lin = timed_graph(lambda: F.linear(A, W))
                rec[f"M_{tag}"] = M
                rec[f"hipblaslt_matmul_us_{tag}"] = round(us_lib, 2)
                rec[f"hipblaslt_linear_us_{tag}"] = round(us_lin, 2)
                if tag == "pad":
                    e = _lib.GemmEpilogue()
                    e.out_bf = out.data_ptr(); e.ldbf = N
                    us_own = timed_graph(lambda: _lib.check(L.splice_gemm_nt_bf16(_lib.EPI_OUT_BF, _lib.ptr(A), K, _lib.ptr(W), K, M, N, K, C.byref(e), _lib.current_stream())))
                    ref = A.float() @ W.float().t()
                    err = ((out.float() - ref).norm() / ref.norm()).item()
                    rec["own_us"] = round(us_own, 2)
                    rec["own_relerr"] = err
            fl = 2.0 * M_pad * N * K
            best_lib = min(rec["hipblaslt_matmul_us_pad"], rec["hipblaslt_linear_us_pad"])
            rec["own_TF"] = round(fl / rec["own_us"] / 1e6, 1)
            rec["lib_TF"] = round(fl / best_lib / 1e6, 1)
            rec["own_over_lib"] = round(rec["own_us"] / best_lib, 3)
            rows.append(rec)
            print(f"gemm {name:6s} M={M_pad:6d} N={N:5d} K={K:5d}: own {rec['own_us']:7.1f} us {rec['own_TF']:6.0f} TF | hipBLASLt matmul {rec['hipblaslt_matmul_us_pad']:7.1f} "
                  f"linear {rec['hipblaslt_linear_us_pad']:7.1f} us {rec['lib_TF']:6.0f} TF (unpadded M={M_raw}: {rec['hipblaslt_matmul_us_raw']:7.1f} / {rec['hipblaslt_linear_us_raw']:7.1f}) "
                  f"| own/lib {rec['own_over_lib']:.2f}", flush=True)
    return rows


def sdpa_backends():
    out = [("default", None)]
    try:
        from torch.nn.attention import SDPBackend, sdpa_kernel
        for nm in ("FLASH_ATTENTION", "EFFICIENT_ATTENTION", "MATH"):
            if hasattr(SDPBackend, nm):
                out.append((nm.lower(), (sdpa_kernel, getattr(SDPBackend, nm))))
    except Exception as ex:   # noqa: BLE001
        print("sdpa_kernel context unavailable:", ex)
    return out


def attn_table():
    rows = []
    D, H, hd = 768, 12, 64
    scale = hd ** -0.5
    for (B, T) in ((4, 785), (32, 785), (4, 3137)):
        Tld = (T + 31) // 32 * 32
        nrows = B * Tld
        g = torch.Generator(device="cpu").manual_seed(5)
        qkv = torch.randn(nrows, 3 * D, generator=g).to(DEV).bfloat16()
        dout = torch.randn(B, Tld, D, generator=g)
        dout[:, T:] = 0
        dout = dout.reshape(nrows, D).to(DEV).bfloat16()
        fl_f = 4.0 * T * T * hd * H * B
        rec = dict(kind="attention", B=B, T=T)
        # --- own kernels, the engine's configuration: q pre-scaled by scale * log2(e) (fold) ---
        qf = qkv.clone()
        qf[:, :D] = (qf[:, :D].float() * (scale * 1.4426950408889634)).bfloat16()
        out = torch.zeros(nrows, D, device=DEV, dtype=torch.bfloat16)
        lse = torch.zeros(B, H, Tld, device=DEV)
        delta = torch.zeros(B, H, Tld, device=DEV)
        dqkv = torch.zeros(nrows, 3 * D, device=DEV, dtype=torch.bfloat16)
        st = _lib.current_stream()
        L.splice_attention_qfold(1)
        ks = 0.6931471805599453

        def fwd():
            _lib.check(L.splice_attention_fwd(_lib.ptr(qf), None, nrows, B, T, Tld, D, H, ks, _lib.ptr(out), _lib.ptr(lse), st))

        def bwd():
            _lib.check(L.splice_attention_bwd(_lib.ptr(qf), None, nrows, B, T, Tld, D, H, ks, _lib.ptr(out), _lib.ptr(lse),
                                              _lib.ptr(dout), None, _lib.ptr(delta), _lib.ptr(dqkv), st))
        fwd(); bwd()
        rec["own_fwd_us"] = round(timed(fwd), 2)
        rec["own_bwd_us"] = round(timed(bwd), 2)
        L.splice_attention_qfold(0)
        # --- SDPA on [B, H, T, hd] (unpadded T: the library's best case) ---
        x = qkv.reshape(B, Tld, 3, H, hd)[:, :T]
        q = x[:, :, 0].transpose(1, 2).contiguous().requires_grad_(True)
        k = x[:, :, 1].transpose(1, 2).contiguous().requires_grad_(True)
        v = x[:, :, 2].transpose(1, 2).contiguous().requires_grad_(True)
        go = dout.reshape(B, Tld, H, hd)[:, :T].transpose(1, 2).contiguous()
        for nm, ctx in sdpa_backends():
            try:
                def run_f():
                    with torch.no_grad():
                        return F.scaled_dot_product_attention(q, k, v)

                def run_fb():
                    o = F.scaled_dot_product_attention(q, k, v)
                    o.backward(go)
                    q.grad = k.grad = v.grad = None
                if ctx is None:
                    tf = min(timed(run_f, n=30, warm=5), timed_graph(run_f, n=10))   # graph replay: without torch's host path (matters at B = 4, T = 785)
                    tfb = timed(run_fb, n=20, warm=3)
                else:
                    with ctx[0](ctx[1]):
                        tf = min(timed(run_f, n=30, warm=5), timed_graph(run_f, n=10))
                        tfb = timed(run_fb, n=20, warm=3)
                rec[f"sdpa_{nm}_fwd_us"] = round(tf, 2)
                rec[f"sdpa_{nm}_fwd_bwd_us"] = round(tfb, 2)
            except Exception as ex:   # noqa: BLE001
                rec[f"sdpa_{nm}_error"] = str(ex).splitlines()[0][:160]
        rec["own_fwd_TF"] = round(fl_f / rec["own_fwd_us"] / 1e6, 1)
        rec["own_bwd_TF_alg2.5x"] = round(2.5 * fl_f / rec["own_bwd_us"] / 1e6, 1)
        rows.append(rec)
        print("attn", json.dumps(rec), flush=True)
    return rows


def copy_rate():
    """The box's f32 copy rate (float4 loads/stores through torch's copy kernel), GB/s moved (read + write)."""
    n = 1 << 28   # 1 GiB of floats... 256 Mi floats = 1 GiB
    a = torch.empty(n, device=DEV, dtype=torch.float32).normal_()
    b = torch.empty_like(a)
    us = timed(lambda: b.copy_(a), n=10, warm=3)
    return 2.0 * n * 4 / us / 1e3


if __name__ == "__main__":
    what = sys.argv[1].split(",") if len(sys.argv) > 1 else ["copy", "gemm", "attn"]
    res = dict(device=torch.cuda.get_device_name(0), torch=torch.__version__, hip=torch.version.hip)
    if "copy" in what:
        res["hbm_copy_GBps"] = round(copy_rate(), 1)
        print("f32 copy rate:", res["hbm_copy_GBps"], "GB/s", flush=True)
    if "gemm" in what:
        res["gemm"] = gemm_table()
    if "attn" in what:
        res["attention"] = attn_table()
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/yardstick.json", "w") as fh:
        json.dump(res, fh, indent=1)
