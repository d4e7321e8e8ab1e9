"""usage (GPU box): python tools/gen_bits.py [N h w]  -- digests of the generator engine's forward output, BatchNorm statistics and
every parameter-gradient tensor on seeded inputs.  Run it under different SPLICE_* launch-form switches and diff the output: forms
that claim to be bit-neutral must print the same lines (tools/gen_bits_ab.sh)."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch

from splice_amd import synth
from splice_amd.generator import GeneratorEngine

N, h, w = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (1, 224, 224)
eng = GeneratorEngine()
raw = synth.generator_params(5, 0.02, perturb_bias=0.03)
params = eng.flatten(raw)
x = torch.from_numpy(synth.uniform(6, "gx", (N, 3, h, w))).cuda()
wgt = torch.from_numpy(synth.normal(7, "gw", (N, 3, h, w))).cuda()
plan = eng.plan(N, h, w, need_grad=True)
y = plan.forward(params, x)
grads = plan.backward(params, wgt)
torch.cuda.synchronize()


def dig(t):
    return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:16]


if os.environ.get("GEN_BITS_DUMP"):   # tensors themselves, for numeric comparison of two builds / switch settings (tools/gen_num_ab.sh)
    torch.save({"forward": y.cpu(), **{k: v.cpu() for k, v in eng.unflatten(grads).items()}}, os.environ["GEN_BITS_DUMP"])
print("forward", dig(y))
for name, g in eng.unflatten(grads).items():
    print(name, dig(g))
