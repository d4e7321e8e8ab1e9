# HBM-side traffic of one kernel per launch: separate --pmc passes (never combined with sys/hip traces), gfx950 FETCH_SIZE x2
# usage: bash tools/pmc_traffic.sh '<kernel name substring>'     e.g. 'gemm_nt_kernel<64, 64, 7u, 4>'
cd /tmp && export TMPDIR=/tmp
K="$1"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$c -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --prof-kernel 0 > /dev/null 2>&1
done
python - "$K" <<'PY'
import csv, glob, os, sys
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = []
    for f in glob.glob(f"{root}/pmc_{c}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if sys.argv[1] in r["Kernel_Name"] and r["Counter_Name"] == c:
                vals.append(float(r["Counter_Value"]))
    out[c] = (sum(vals) / max(len(vals), 1), len(vals))
f, w = out["FETCH_SIZE"][0], out["WRITE_SIZE"][0]
print(f"kernel '{sys.argv[1]}': FETCH_SIZE {f:.1f} KB x2 (gfx950) + WRITE_SIZE {w:.1f} KB over {out['FETCH_SIZE'][1]} launches -> {(2*f+w)*1024/1e6:.2f} MB per launch (rocprofv3 reports KB)")
PY
