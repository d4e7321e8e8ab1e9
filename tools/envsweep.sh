run() { echo -n "$* -> "; env "$@" timeout 150 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --prof-kernels '' --pairs-sweep '' --no-train-regime 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null || echo FAIL; }
run X=1
run HIP_FORCE_DEV_KERNARG=0
run HIP_FORCE_DEV_KERNARG=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run DEBUG_HIP_FORCE_GRAPH_QUEUES=2
run DEBUG_HIP_FORCE_GRAPH_QUEUES=4
run AMD_OPT_FLUSH=0
run AMD_OPT_FLUSH=1
run ROC_SYSTEM_SCOPE_SIGNAL=0
run GPU_STREAMOPS_CP_WAIT=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=64
run DEBUG_HIP_GRAPH_BATCH_SIZE=1024
run DEBUG_HIP_DYNAMIC_QUEUES=0
run DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0
run X=2
