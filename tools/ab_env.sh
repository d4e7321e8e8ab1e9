# usage (GPU box): bash tools/ab_env.sh "<bench args>" VAR=a VAR=b [...]   -- alternating same-box A/B/A/B/A/B of environment settings
ARGS="$1"; shift
for rep in 1 2 3; do
  for kv in "$@"; do
    v=$(env $kv python bench.py $ARGS --no-cpu-baseline --no-train-regime --pairs-sweep "" --prof-kernels "" 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "$kv $v"
  done
done
