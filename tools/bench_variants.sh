# usage (GPU box): bash tools/bench_variants.sh   -- the side rows of DESIGN.md section 8 (other sizes / models / modes), one line each
run() { echo -n "$* -> "; python bench.py "$@" --steps 100 --warmup 15 --no-cpu-baseline --prof-kernels "" --pairs-sweep "" --no-train-regime 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], 'steps/s', j['ms_per_step'], 'ms')"; }
run --size 448
run --size 320
run --model dino_vitb16
run --model dino_vits8
run --scales 224,320,448
run --fp8
run
