# usage (GPU box): bash tools/multiproc_gpu.sh   -- W worker processes sharing ONE GPU, each optimising P pairs per step:
# aggregate pair-steps/s (do independent processes fill the gaps a single step leaves?)
run() {  # W P
  W=$1; P=$2; rm -f /tmp/mp_*.json
  for w in $(seq $W); do python bench.py --pairs $P --steps 120 --warmup 30 --no-cpu-baseline --pairs-sweep "" --no-train-regime --prof-kernels "" > /tmp/mp_$w.json 2>/dev/null & done
  wait
  python - "$W" "$P" <<'PY'
import json, sys, glob
W, P = int(sys.argv[1]), int(sys.argv[2])
vals = [json.loads(open(f).read().strip().splitlines()[-1])["config"]["pair_steps_per_s"] for f in sorted(glob.glob("/tmp/mp_*.json"))]
print(f"{W} process(es) x {P} pair(s) per step: per process {[round(v, 1) for v in vals]} -> {sum(vals):.1f} pair-steps/s on the GPU", flush=True)
PY
}
run 1 1; run 2 1; run 3 1; run 4 1; run 1 4; run 2 4; run 1 8; run 2 8; run 4 2
