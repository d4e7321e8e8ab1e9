#!/bin/bash
# A/B of the Q-fold experiment (-DATTN_QFOLD: q pre-multiplied by scale*log2(e) and re-rounded to bf16 in registers, score accumulators
# seeded with -m, p = exp2(acc) -- no per-score FMA) against the shipped forward.  Build here: see profiles/r04_attn_qfold_experiment.txt
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cp $ROOT/splice_amd/libsplice_hip.so /tmp/keep_qf.so
for r in 1 2 3; do for L in head qfold; do
  cp $ROOT/build/abl/lib_$L.so $ROOT/splice_amd/libsplice_hip.so
  echo "== $L (round $r)"
  ATTN_SHAPES=${1:-16x785,2x785,2x3137} python $ROOT/tools/attn_bench.py ${2:-0} 2>&1 | grep -v amdgpu.ids | sed 's/| bwd.*TF(alg 2.5x)//'
done; done
cp /tmp/keep_qf.so $ROOT/splice_amd/libsplice_hip.so
