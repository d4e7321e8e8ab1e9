timeout 1500 python -m pytest tests -m gpu -q --durations=5 2>&1 | tail -15 > gpurun_out/r03_c_tests.log
timeout 400 python bench.py > gpurun_out/r03_c_bench.json 2> gpurun_out/r03_c_bench.err
bash tools/prof_step.sh r03_p1 --steps 60 --warmup 10 > /dev/null 2>&1
bash tools/prof_step.sh r03_p8 --pairs 8 --steps 25 --warmup 10 > /dev/null 2>&1
bash tools/pmc_selfsim.sh 1 > gpurun_out/r03_pmc_attn_selfsim_p1.txt 2>&1
cat gpurun_out/r03_c_tests.log; tail -c 400 gpurun_out/r03_c_bench.json
