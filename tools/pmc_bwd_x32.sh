cd /tmp && export TMPDIR=/tmp
ATTN_FOLD=1 SPLICE_ATTN_BWD_VARIANT=3 ATTN_SHAPES=4x3137 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_bwd -- python $GRAFT_REPO_ROOT/tools/attn_bench.py 41 > /dev/null 2>&1
python - <<'PY'
import csv, glob, os
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof_bwd"
for f in glob.glob(root + "/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "attn" in r["Name"]: print(r["Name"][:60], r["Calls"], r["AverageNs"])
PY
ATTN_FOLD=1 SPLICE_ATTN_BWD_VARIANT=3 bash $GRAFT_REPO_ROOT/tools/pmc_attn.sh 41 4x3137 2>&1 | grep -A27 'bwd_q_x32\|bwd_kv_x32' | grep -E '==|BANK_CONFLICT|LDS_IDX_ACTIVE|WAVE_CYCLES|WAIT_ANY|WAIT_INST_ANY|ACTIVE_INST_ANY|INSTS_|MFMA_BUSY|GRBM_GUI'
