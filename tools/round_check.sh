# usage (GPU box): bash tools/round_check.sh [tag]  -- the whole -m gpu suite, the default bench, the fp8 / multi-scale bench lines,
# rocprofv3 kernel statistics at 1 and 8 pairs per GPU and the SQ counters of the attention / self-similarity kernels
TAG=${1:-r03}
timeout 1500 python -m pytest tests -m gpu -q --durations=5 2>&1 | tail -15 > gpurun_out/${TAG}_tests.log
timeout 400 python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
timeout 300 python bench.py --fp8 attention --no-cpu-baseline --no-train-regime --prof-kernels 4,1,2,3 > gpurun_out/${TAG}_bench_fp8.json 2> gpurun_out/${TAG}_bench_fp8.err
timeout 300 python bench.py --scales 224,320,448 --fp8 attention --steps 60 --warmup 10 > gpurun_out/${TAG}_bench_scales_fp8.json 2> gpurun_out/${TAG}_bench_scales_fp8.err
timeout 300 python bench.py --scales 224,320,448 --steps 60 --warmup 10 > gpurun_out/${TAG}_bench_scales.json 2> gpurun_out/${TAG}_bench_scales.err
timeout 300 python bench.py --scales 224,320,448 --fp8 gemm --steps 60 --warmup 10 > gpurun_out/${TAG}_bench_scales_fp8_gemm.json 2> gpurun_out/${TAG}_bench_scales_fp8_gemm.err
timeout 300 python bench.py --fp8 gemm --no-cpu-baseline --no-train-regime --prof-kernels 4,1,2,3 > gpurun_out/${TAG}_bench_fp8_gemm.json 2> gpurun_out/${TAG}_bench_fp8_gemm.err
timeout 300 python bench.py --pairs 8 --steps 60 --warmup 10 --no-cpu-baseline --no-train-regime --pairs-sweep "" > gpurun_out/${TAG}_bench_pairs8.json 2> gpurun_out/${TAG}_bench_pairs8.err
timeout 300 python bench.py --pairs 8 --fp8 gemm --steps 60 --warmup 10 --no-cpu-baseline --no-train-regime --pairs-sweep "" > gpurun_out/${TAG}_bench_pairs8_fp8_gemm.json 2> gpurun_out/${TAG}_bench_pairs8_fp8_gemm.err
bash tools/prof_step.sh ${TAG}_p1 --steps 60 --warmup 10 > /dev/null 2>&1
bash tools/prof_step.sh ${TAG}_p8 --pairs 8 --steps 25 --warmup 10 > /dev/null 2>&1
bash tools/pmc_selfsim.sh 1 > gpurun_out/${TAG}_pmc_attn_selfsim_p1.txt 2>&1
cat gpurun_out/${TAG}_tests.log; tail -c 300 gpurun_out/${TAG}_bench_default.json; tail -c 300 gpurun_out/${TAG}_bench_scales_fp8.json
