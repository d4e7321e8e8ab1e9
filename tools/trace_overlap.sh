# kernel-trace of a short bench run; reports, for the last steps, wall time vs union of kernel intervals vs sum of durations
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/trace
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --prof-kernels '' --pairs-sweep '' --no-train-regime > /dev/null 2>&1
python - <<'PY'
import csv, glob, os
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/trace/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:50], r.get("Queue_Id", "")) for r in rows)
# find adam kernels = step boundaries
adam = [i for i, e in enumerate(ev) if "adam" in e[2]]
print("kernels", len(ev), "adam launches", len(adam))
for a, b in list(zip(adam[:-1], adam[1:]))[-6:-1]:
    seg = ev[a + 1:b + 1]
    wall = seg[-1][1] - ev[a][1]
    ssum = sum(e[1] - e[0] for e in seg)
    # union
    cur_s, cur_e, uni = seg[0][0], seg[0][1], 0
    for s_, e_, *_ in seg[1:]:
        if s_ > cur_e: uni += cur_e - cur_s; cur_s, cur_e = s_, e_
        else: cur_e = max(cur_e, e_)
    uni += cur_e - cur_s
    qs = sorted(set(e[3] for e in seg))
    print(f"step: {len(seg)} kernels, wall {wall/1e3:.0f} us, union busy {uni/1e3:.0f} us, sum of durations {ssum/1e3:.0f} us, queues {qs}")
PY
