#!/bin/bash
# dump the gfx950 ISA of one translation unit: tools/isa.sh attention  -> /tmp/isa/attention.s
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -mllvm -amdgpu-kernarg-preload-count=16 -I/root/repo/include -S --cuda-device-only -o /tmp/isa/$1.s /root/repo/splice_amd/csrc/$1.hip 2>&1 | grep -v "hip-link"
python3 - "$1" <<'PY'
import re, sys
s = open(f"/tmp/isa/{sys.argv[1]}.s").read()
meta = s[s.index("amdhsa.kernels"):]
for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)", meta):
    name = m.group(1)
    try:
        i = s.index(name + ":"); j = s.index(".Lfunc_end", i); body = s[i:j]
    except ValueError:
        continue
    print(f"{name[:60]:60s} vgpr {m.group(2):>4s} lines {body.count(chr(10)):5d} mfma {body.count('v_mfma'):3d} exp {body.count('v_exp'):3d} "
          f"vmcnt {body.count('vmcnt'):2d} barrier {body.count('s_barrier'):2d} scratch {body.count('scratch_'):2d} v_mov {body.count('v_mov'):3d}")
PY
