# HBM-side traffic per launch of the kernels whose name contains a substring: separate --pmc passes (never combined with
# sys/hip traces), gfx950 FETCH_SIZE x2 (MI355X_MICROARCH.md, HBM section).
# usage: bash tools/pmc_traffic.sh <tag> '<kernel name substring>' [bench args...]    e.g.  p1 'gemm_nt_kernel<64, 64' --pairs 1
# Writes gpurun_out/traffic_<tag>.txt.
TAG=$1; K="$2"; shift; shift
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_${TAG}_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_${TAG}_$c -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --prof-kernels "" --pairs-sweep "" --no-train-regime "$@" > /dev/null 2>&1
done
python - "$K" "$TAG" <<'PY' | tee $GRAFT_REPO_ROOT/gpurun_out/traffic_$TAG.txt
import csv, glob, sys
# One template instantiation can serve several shapes (a 64x64 ring kernel runs both fc2 forward, K = 3072, and proj
# forward, K = 768): per full kernel name the launches are clustered by FETCH_SIZE, every cluster with its own averages
# (launch order is the same in both passes, so the i-th launch of a name pairs its FETCH with its WRITE).
sub, tag = sys.argv[1], sys.argv[2]
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = []
    for f in glob.glob(f"/tmp/pmc_{tag}_{c}/*/*counter_collection.csv"):
        rows += [r for r in csv.DictReader(open(f)) if sub in r["Kernel_Name"] and r["Counter_Name"] == c]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    for r in rows:
        vals.setdefault(r["Kernel_Name"], {}).setdefault(c, []).append(float(r["Counter_Value"]))
for name, d in sorted(vals.items()):
    f, w = d.get("FETCH_SIZE", []), d.get("WRITE_SIZE", [])
    n = min(len(f), len(w))
    if n == 0:
        continue
    pairs = sorted(zip(f[:n], w[:n]))
    clusters, cur = [], [pairs[0]]
    for p in pairs[1:]:
        if p[0] > 1.25 * cur[0][0] + 64:
            clusters.append(cur); cur = [p]
        else:
            cur.append(p)
    clusters.append(cur)
    print(name[:150])
    for cl in clusters:
        fa, wa = sum(p[0] for p in cl) / len(cl), sum(p[1] for p in cl) / len(cl)
        print(f"    {len(cl):5d} launches: FETCH_SIZE {fa:9.1f} KB x2 (gfx950) + WRITE_SIZE {wa:9.1f} KB -> {(2*fa+wa)*1024/1e6:8.2f} MB per launch")
PY
