import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from splice_amd import _lib
L = _lib.lib()
B, T, D, H = 2, 785, 768, 12
Tld = (T + 31) // 32 * 32
rows = B * Tld
g = torch.Generator().manual_seed(77)
x = torch.randn(rows, 3 * D, generator=g)
spikes = [(0, 5, 3, 300), (1, 700, 11, 784), (1, 64, 0, 65)]
for (b, q, h, key) in spikes:
    x[b * Tld + q, h * 64:(h + 1) * 64] = 40.0
    x[b * Tld + key, D + h * 64:D + (h + 1) * 64] = 40.0
qkv = x.cuda().bfloat16()
for variant in (1, 41, 42):
    L.splice_attention_variant(variant)
    out = torch.zeros(rows, D, device="cuda", dtype=torch.bfloat16)
    lse = torch.zeros(B, H, Tld, device="cuda")
    _lib.check(L.splice_attention_fwd(_lib.ptr(qkv), None, 0, B, T, Tld, D, H, 0.125, _lib.ptr(out), _lib.ptr(lse), _lib.current_stream()))
    torch.cuda.synchronize()
    o = out.float().reshape(B, Tld, H, 64)
    bad = (~torch.isfinite(o)).any(-1).nonzero()
    print("variant", variant, "non-finite (b, q, h):", bad[:20].tolist(), "count", len(bad), "lse bad", (~torch.isfinite(lse)).nonzero()[:10].tolist())
L.splice_attention_variant(0)
