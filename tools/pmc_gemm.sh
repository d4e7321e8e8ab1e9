# usage: bash tools/pmc_gemm.sh M N K tile   (PMC passes only)
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_INSTS_SMEM SQ_WAVES_EQ_64 SQ_INST_LEVEL_LDS" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_gemm/p$i -- python $GRAFT_REPO_ROOT/tools/gemm_one.py $1 $2 $3 $4 20 > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, collections, os
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_gemm"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm" not in k: continue
        acc[k.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print("==", k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} {sum(v)/len(v):14.0f}  (n={len(v)})")
PY
