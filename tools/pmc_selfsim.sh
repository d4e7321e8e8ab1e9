# usage: bash tools/pmc_selfsim.sh <pairs> [kernel name substrings, comma separated; default "selfsim,attn_"] [more bench args, e.g. --size 448]
#   -- SQ / GRBM / L2 counters of kernels inside bench.py steps (PMC passes only, eager single-stream launches)
cd /tmp && export TMPDIR=/tmp
P=${1:-1}
MATCH=${2:-selfsim,attn_}
shift; shift
TAGX=$(echo "$@" | tr -d ' -')
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  SPLICE_STEP_GRAPH=0 SPLICE_STEP_OVERLAP=0 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_ss_$P$TAGX/p$i -- python $GRAFT_REPO_ROOT/bench.py --pairs $P --steps 6 --warmup 2 --no-cpu-baseline --prof-kernels '' --pairs-sweep '' --no-train-regime --allow-dev-env "$@" > /dev/null 2>&1
done
PMC_ROOT=/tmp/pmc_ss_$P$TAGX PMC_MATCH="$MATCH" python - <<'PY'
import csv, glob, collections, os
root = os.environ["PMC_ROOT"]; match = os.environ["PMC_MATCH"].split(",")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not any(m in k for m in match): continue
        acc[k.split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] == "SQ_WAVES":   # grid size separates the shapes one instantiation serves
            acc[k.split("(")[0].replace("void ", "")]["(launches by SQ_WAVES: " + r["Counter_Value"].split(".")[0] + ")"].append(1.0)
for k, d in sorted(acc.items()):
    print("==", k)
    for c, v in sorted(d.items()):
        print(f"   {c:36s} {sum(v)/len(v):14.0f}  (n={len(v)})")
    g = lambda n: (sum(d[n]) / len(d[n])) if n in d else 0.0
    if g("TCC_HIT_sum") + g("TCC_MISS_sum") > 0:
        print(f"   -> L2 hit rate = {g('TCC_HIT_sum') / (g('TCC_HIT_sum') + g('TCC_MISS_sum')):.3f}")
    if g("SQ_INSTS_MFMA"):
        print(f"   -> SQ_INSTS_VALU / SQ_INSTS_MFMA = {g('SQ_INSTS_VALU') / g('SQ_INSTS_MFMA'):.2f}; SQ_INSTS_SALU / SQ_INSTS_MFMA = {g('SQ_INSTS_SALU') / g('SQ_INSTS_MFMA'):.2f}")
    if g("SQ_WAVE_CYCLES"):
        # counters are summed over the 8 XCDs; GRBM_GUI_ACTIVE / 8 = kernel cycles (under the profiler); 1024 SIMDs
        util = g('SQ_VALU_MFMA_BUSY_CYCLES') / max(g('GRBM_GUI_ACTIVE') / 8 * 1024, 1) if g('GRBM_GUI_ACTIVE') else float('nan')
        print(f"   -> WAIT_ANY / WAVE_CYCLES = {g('SQ_WAIT_ANY')/g('SQ_WAVE_CYCLES'):.2f}; MFMA busy SIMD-cycles / (kernel cycles x 1024 SIMDs) = {util:.3f} (profiled run); LDS conflict / LDS active = {g('SQ_LDS_BANK_CONFLICT')/max(g('SQ_LDS_IDX_ACTIVE'),1):.3f}")
PY
