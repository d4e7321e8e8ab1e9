#!/usr/bin/env python3
"""Per kernel: global loads vs vmcnt waits in the gfx950 ISA (tools/isa.sh <unit> first).  A ratio near 1 means the
loads are serialised (guarded loads each followed by a wait): python tools/loadwait.py gen_kernels"""
import re, sys
s = open(f"/tmp/isa/{sys.argv[1]}.s").read()
for m in re.finditer(r"^(_Z\w+):.*\n", s, re.M):
    name = m.group(1)
    j = s.find(".Lfunc_end", m.end())
    body = s[m.end():j]
    nl = len(re.findall(r"\b(global_load|buffer_load)", body)); nw = len(re.findall(r"vmcnt\(", body))
    nb = body.count("s_cbranch")
    if nl: print(f"{name[:70]:70s} loads {nl:4d} vmcnt-waits {nw:4d} branches {nb:4d}")
