# PMC profile of any kernel inside the benchmark step (counters only, never combined with sys/hip traces):
# usage: bash tools/pmc_kernel.sh '<kernel name substring>'     e.g. conv_wgrad_batched
cd /tmp && export TMPDIR=/tmp
K="$1"
i=0
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_k
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  SPLICE_STEP_OVERLAP=0 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_k/p$i -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --prof-kernels '' --pairs-sweep '' --no-train-regime --allow-dev-env > /dev/null 2>&1
done
python - "$K" <<'PY'
import csv, glob, collections, os, sys
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_k"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if sys.argv[1] not in k: continue
        acc[k.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print("==", k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} {sum(v)/len(v):14.0f}  (n={len(v)})")
PY
