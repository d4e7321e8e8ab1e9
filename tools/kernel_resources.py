#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table from hipcc's -Rpass-analysis (no GPU needed).
usage: python tools/kernel_resources.py splice_amd/csrc/gemm.hip [...]"""
import re
import subprocess
import sys

ROOT = __file__.rsplit("/tools/", 1)[0]
for src in sys.argv[1:]:
    out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-mfma-vgpr-form=1", f"-I{ROOT}/include",
                          f"-I{ROOT}/splice_amd/csrc", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"],
                         capture_output=True, text=True).stderr
    cur = {}
    rows = []
    for line in out.splitlines():
        m = re.search(r"remark:\s+(.*?):\s+(\S+)\s+\[-Rpass", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2)
        if k == "Function Name":
            cur = {"name": v}
            rows.append(cur)
        else:
            cur[k] = v
    print(f"== {src}")
    for r in rows:
        name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name)[:70]
        print(f"{name:70s} vgpr {r.get('VGPRs','?'):>4} agpr {r.get('AGPRs','?'):>4} scratch {r.get('ScratchSize [bytes/lane]','?'):>5} "
              f"occ {r.get('Occupancy [waves/SIMD]','?'):>2} lds {r.get('LDS Size [bytes/block]','?'):>6}")
