#!/bin/bash
# usage (GPU box): bash tools/critical_path_map.sh <dev_lib.so> [pairs] -- what each launch region costs ON THE CRITICAL PATH of the step (round 6).
# The scratch library (make DEV=1 OUT=...) does not issue the launches of a region whose bit is set in SPLICE_DEV_SKIP once two steps have run:
# TIMING ONLY, the results are garbage (stale but finite buffers).  Step time without a region against the unmodified step = the region's cost where
# it matters; the serialised kernel time of rocprofv3 says what a region costs when nothing overlaps it.  bench.py marks such lines (config.dev_env).
LIB=$1; P=${2:-1}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cp $ROOT/splice_amd/libsplice_hip.so /tmp/keep_cpm.so
cp $LIB $ROOT/splice_amd/libsplice_hip.so
names=(none LN_fwd QKV_fwd attn_fwd proj_fwd fc1_fwd fc2_fwd fc2T fc1T LN_bwd projT attn_bwd qkvT BN_fwd BN_bwd conv_fwd_rest conv_dgrad_rest wgrad_tail selfsim cls_tail_fwd cls_tail_bwd adam)
run() { SPLICE_DEV_SKIP=$1 python $ROOT/bench.py --pairs $P --steps 100 --warmup 10 --no-cpu-baseline --pairs-sweep "" --no-train-regime --prof-kernels "" --allow-dev-env 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
base=$(run 0)
echo "pairs per GPU $P: unmodified step $base ms"
for r in $(seq 1 21); do
  v=$(run $((1 << r)))
  python - "$base" "$v" "${names[$r]}" <<'PY'
import sys
b, v, n = float(sys.argv[1]), float(sys.argv[2]), sys.argv[3]
print(f"  without {n:16s} {v:8.4f} ms  ({(b - v) * 1e3:+7.1f} us, {100 * (b - v) / b:+5.2f} % of the step)")
PY
done
base2=$(run 0)
echo "unmodified step again: $base2 ms"
cp /tmp/keep_cpm.so $ROOT/splice_amd/libsplice_hip.so
