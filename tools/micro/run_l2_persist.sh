# usage (GPU box): bash tools/micro/run_l2_persist.sh
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -o /tmp/l2_persist tools/micro/l2_persist.hip || exit 1
for MODE in 0 1 2; do
echo "allocation mode $MODE (0 hipMalloc, 1 fine-grained, 2 uncached)"
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/l2p && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/l2p -- /tmp/l2_persist $MODE > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob("/tmp/l2p/*/*counter_collection.csv"):
    rows += [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == "FETCH_SIZE"]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
for r in rows:
    print(f'dispatch {r["Dispatch_Id"]:>3s} {r["Kernel_Name"][:40]:40s} FETCH_SIZE {float(r["Counter_Value"]):10.1f} KB (x2 on gfx950; the region is 8192 KB)')
PY
done
