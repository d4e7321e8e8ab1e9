cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/gtrace
SPLICE_STEP_ABLATE=${ABL:-30} rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/gtrace -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --prof-kernels '' --pairs-sweep '' --no-train-regime --allow-dev-env > /dev/null 2>&1
python - <<'PY'
import csv, glob, os
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/gtrace/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ","")[:48], r["Grid_Size_X"], r["Workgroup_Size_X"]) for r in rows)
adam = [i for i, e in enumerate(ev) if "adam" in e[2]]
a, b = adam[-3], adam[-2]
seg = ev[a+1:b+1]
t0 = seg[0][0]
tot_d = tot_g = 0
prev = None
for s_, e_, n, gx, wx in seg:
    gap = (s_ - prev) / 1e3 if prev else 0
    tot_d += (e_ - s_) / 1e3; tot_g += max(gap, 0)
    print(f"{(s_-t0)/1e3:8.1f} dur {(e_-s_)/1e3:6.1f} gap {gap:5.1f} grid {int(gx)//int(wx):6d}  {n}")
    prev = e_
print("sum dur", tot_d, "sum gaps", tot_g, "wall", (seg[-1][1]-t0)/1e3)
PY
