#!/bin/bash
# usage (GPU box): bash tools/gen_num_ab.sh VAR=a VAR=b ["N h w" ...]  -- the generator's output and every gradient tensor under two settings of an environment
# switch that is NOT bit-neutral (another summation order): largest relative L2 difference over the tensors, and which tensor.
A=$1; B=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for sz in "${@:-1 224 224}"; do
  env $A GEN_BITS_DUMP=/tmp/num_a.pt python $ROOT/tools/gen_bits.py $sz > /dev/null 2>/tmp/num_a.err || tail -3 /tmp/num_a.err
  env $B GEN_BITS_DUMP=/tmp/num_b.pt python $ROOT/tools/gen_bits.py $sz > /dev/null 2>/tmp/num_b.err || tail -3 /tmp/num_b.err
  python - "$sz" "$A" "$B" <<'PY'
import sys, torch
a, b = torch.load("/tmp/num_a.pt"), torch.load("/tmp/num_b.pt")
worst, name, ndiff = 0.0, None, 0
for k in a:
    d = (a[k].double() - b[k].double()).norm().item(); n = a[k].double().norm().item()
    r = d / max(n, 1e-30)
    ndiff += d > 0
    if r > worst: worst, name = r, k
print(f"size [{sys.argv[1]}] {sys.argv[2]} | {sys.argv[3]}: {ndiff} of {len(a)} tensors differ, worst relative L2 {worst:.3e} ({name})")
PY
done
