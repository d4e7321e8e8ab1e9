# usage (GPU box): bash tools/round_check.sh [tag]  -- the whole -m gpu suite, the default / 448 / 8-pair / fp8 / multi-scale bench lines,
# rocprofv3 kernel statistics at 1 and 8 pairs per GPU, per-family HBM-side traffic (separate --pmc passes) and the SQ counters of the
# attention / self-similarity kernels at 224 and 448.  Everything lands in gpurun_out/<tag>_*; copy what is to be kept into profiles/.
TAG=${1:-r06}
B="--no-cpu-baseline --no-train-regime --pairs-sweep \"\""
timeout 1500 python -m pytest tests -m gpu -q --durations=5 2>&1 | tail -15 > gpurun_out/${TAG}_gpu_tests.log
bash tools/pmc_families.sh ${TAG}_p1 > /dev/null 2>&1
bash tools/pmc_families.sh ${TAG}_p8 --pairs 8 > /dev/null 2>&1
bash tools/pmc_families.sh ${TAG}_s448 --size 448 > /dev/null 2>&1
bash tools/pmc_families.sh ${TAG}_s512 --size 512 > /dev/null 2>&1
bash tools/pmc_families.sh ${TAG}_img900 --image 900x1200 > /dev/null 2>&1
python tools/merge_traffic.py ${TAG} P1=gpurun_out/traffic_families_${TAG}_p1.json P8=gpurun_out/traffic_families_${TAG}_p8.json S448P1=gpurun_out/traffic_families_${TAG}_s448.json S512P1=gpurun_out/traffic_families_${TAG}_s512.json IMG900x1200=gpurun_out/traffic_families_${TAG}_img900.json > /dev/null 2>&1; cp profiles/roofline_traffic.json gpurun_out/${TAG}_roofline_traffic.json
timeout 500 python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver_form.json 2> gpurun_out/${TAG}_bench_driver_form.err
timeout 300 python bench.py --size 448 --steps 60 --warmup 10 --no-cpu-baseline --no-train-regime --pairs-sweep "" > gpurun_out/${TAG}_bench_448.json 2> gpurun_out/${TAG}_bench_448.err
timeout 300 python bench.py --pairs 8 --steps 60 --warmup 10 --no-cpu-baseline --no-train-regime --pairs-sweep "" > gpurun_out/${TAG}_bench_pairs8.json 2> gpurun_out/${TAG}_bench_pairs8.err
timeout 300 python bench.py --fp8 gemm --no-cpu-baseline --no-train-regime > gpurun_out/${TAG}_bench_fp8_gemm.json 2> gpurun_out/${TAG}_bench_fp8_gemm.err
timeout 300 python bench.py --pairs 8 --fp8 gemm --steps 60 --warmup 10 --no-cpu-baseline --no-train-regime --pairs-sweep "" > gpurun_out/${TAG}_bench_pairs8_fp8_gemm.json 2> gpurun_out/${TAG}_bench_pairs8_fp8_gemm.err
timeout 300 python bench.py --size 512 --steps 60 --warmup 10 --no-cpu-baseline --no-train-regime --pairs-sweep "" > gpurun_out/${TAG}_bench_512.json 2> gpurun_out/${TAG}_bench_512.err
timeout 300 python bench.py --image 900x1200 --steps 30 --warmup 5 --no-cpu-baseline --pairs-sweep "" > gpurun_out/${TAG}_bench_900_1200.json 2> gpurun_out/${TAG}_bench_900_1200.err
timeout 300 python bench.py --scales 224,320,448 --steps 60 --warmup 10 > gpurun_out/${TAG}_bench_scales.json 2> gpurun_out/${TAG}_bench_scales.err
timeout 300 python bench.py --scales 224,320,448 --fp8 gemm --steps 60 --warmup 10 > gpurun_out/${TAG}_bench_scales_fp8_gemm.json 2> gpurun_out/${TAG}_bench_scales_fp8_gemm.err
bash tools/prof_step.sh ${TAG}_p1 --steps 60 --warmup 10 > /dev/null 2>&1
bash tools/prof_step.sh ${TAG}_p8 --pairs 8 --steps 25 --warmup 10 > /dev/null 2>&1
bash tools/prof_step.sh ${TAG}_s448 --size 448 --steps 30 --warmup 5 > /dev/null 2>&1
bash tools/prof_step.sh ${TAG}_img900 --image 900x1200 --steps 30 --warmup 5 > /dev/null 2>&1
bash tools/pmc_selfsim.sh 1 > gpurun_out/${TAG}_pmc_attn_selfsim_p1.txt 2>&1
bash tools/pmc_selfsim.sh 1 "selfsim,attn_" --size 448 > gpurun_out/${TAG}_pmc_attn_selfsim_448.txt 2>&1
cat gpurun_out/${TAG}_gpu_tests.log; for f in default driver_form 448 512 900_1200 pairs8 fp8_gemm pairs8_fp8_gemm scales scales_fp8_gemm; do python - $f gpurun_out/${TAG}_bench_$f.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], d["value"], d["ms_per_step"], d["config"].get("pair_steps_per_s"), (d.get("roofline") or {}).get("frac"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
