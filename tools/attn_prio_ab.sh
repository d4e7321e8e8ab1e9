#!/bin/bash
# A/B of the s_setprio experiment (-DATTN_PRIO=1: priority 1 around the forward's two MFMA clusters; 2: the backward's too) against the
# same source without it (ATTN_PRIO=0); libraries built under build/abl (see profiles/r04_attn_prio_experiment.txt).
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cp $ROOT/splice_amd/libsplice_hip.so /tmp/keep_pr.so
for r in 1 2 3; do for L in prio0 prio1 prio2; do
  cp $ROOT/build/abl/lib_$L.so $ROOT/splice_amd/libsplice_hip.so
  echo "== $L (round $r)"
  ATTN_SHAPES=${1:-16x785,2x785,2x3137} python $ROOT/tools/attn_bench.py ${2:-0} 2>&1 | grep -v amdgpu.ids | sed 's/ err fwd.*//'
done; done
cp /tmp/keep_pr.so $ROOT/splice_amd/libsplice_hip.so
