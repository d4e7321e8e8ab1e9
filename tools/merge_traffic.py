"""usage: python tools/merge_traffic.py <round tag> P1=gpurun_out/traffic_families_p1.json P8=gpurun_out/traffic_families_p8.json
Rewrites profiles/roofline_traffic.json from tools/pmc_families.sh outputs (the file bench.py reads `roofline.traffic` from)."""
import json
import os
import sys

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
tag = sys.argv[1]
out = {}
for arg in sys.argv[2:]:
    key, path = arg.split("=", 1)
    d = json.load(open(path))
    out[key] = {k: v for k, v in d.items() if not k.startswith("_") and v is not None}
    out[key]["_note"] = (f"bytes per CALL of each live-timed kernel family (ids of bench.kernel_families), {tag}: FETCH_SIZE x2 (gfx950 correction) + "
                         f"WRITE_SIZE from separate --pmc passes of the final build inside the benchmark step (tools/pmc_families.sh; "
                         f"copy of {os.path.basename(path)} under profiles/).  The counters sit on the L2's fabric side and include Infinity-Cache hits.")
    out[key]["_kernels"] = d.get("_detail", {})
json.dump(out, open(os.path.join(root, "profiles", "roofline_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
