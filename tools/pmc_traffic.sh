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
# One template instantiation can serve several shapes (the 64x64 ring kernel runs both fc2 forward, K = 3072, and proj
# forward, K = 768): launches are clustered by their FETCH_SIZE and every cluster is reported with its own average.
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"
rows = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{root}/pmc_{c}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if sys.argv[1] in r["Kernel_Name"] and r["Counter_Name"] == c:
                rows.setdefault(c, []).append(float(r["Counter_Value"]))
f, w = sorted(rows.get("FETCH_SIZE", [0.0])), rows.get("WRITE_SIZE", [0.0])
wavg = sum(w) / len(w)
clusters, cur = [], [f[0]]
for v in f[1:]:
    if v > 1.25 * cur[0]:
        clusters.append(cur); cur = [v]
    else:
        cur.append(v)
clusters.append(cur)
for cl in clusters:
    fa = sum(cl) / len(cl)
    print(f"kernel '{sys.argv[1]}': {len(cl)} launches with FETCH_SIZE {fa:.1f} KB x2 (gfx950) + WRITE_SIZE {wavg:.1f} KB -> {(2*fa+wavg)*1024/1e6:.2f} MB per launch (rocprofv3 reports KB)")
PY
