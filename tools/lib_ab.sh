#!/bin/bash
# usage (GPU box): bash tools/lib_ab.sh "<bench args>" <libA.so> <libB.so> [reps]  -- same-box alternating A/B/A/B of two BUILDS of the library on one bench line
# (the libraries are copied over splice_amd/libsplice_hip.so in turn; the original is restored at the end).  Prints ms per step of every run.
ARGS="$1"; A=$2; B=$3; REPS=${4:-3}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cp $ROOT/splice_amd/libsplice_hip.so /tmp/keep_lib_ab.so
for rep in $(seq 1 $REPS); do
  for L in $A $B; do
    cp $L $ROOT/splice_amd/libsplice_hip.so
    v=$(python $ROOT/bench.py $ARGS --no-cpu-baseline --no-train-regime --pairs-sweep "" --prof-kernels "" 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "$(basename $L) [$ARGS] $v"
  done
done
cp /tmp/keep_lib_ab.so $ROOT/splice_amd/libsplice_hip.so
