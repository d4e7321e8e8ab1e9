# HBM-side traffic per CALL of every kernel family that bench.py's roofline leg times live (ids of bench.kernel_families): separate
# --pmc passes (FETCH_SIZE, WRITE_SIZE; never combined with sys/hip traces), gfx950 FETCH_SIZE x2 (MI355X_MICROARCH.md, HBM section).
# usage (GPU box): bash tools/pmc_families.sh <tag> [bench args...]    e.g.  p8 --pairs 8
# Writes gpurun_out/traffic_families_<tag>.json  ({"<family id>": bytes per call, "_detail": {...}}); copy the numbers into
# profiles/roofline_traffic.json (tools/merge_traffic.py does).
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcf_${TAG}_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmcf_${TAG}_$c -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 4 --no-cpu-baseline --prof-kernels "" --pairs-sweep "" --no-train-regime "$@" > /dev/null 2>&1
done
python - "$TAG" <<'PY' | tee $GRAFT_REPO_ROOT/gpurun_out/traffic_families_$TAG.json
import csv, glob, json, sys
tag = sys.argv[1]
per = {}          # kernel name -> {counter: [values in dispatch order]}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = []
    for f in glob.glob(f"/tmp/pmcf_{tag}_{c}/*/*counter_collection.csv"):
        rows += [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == c]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    for r in rows:
        per.setdefault(r["Kernel_Name"], {}).setdefault(c, []).append(float(r["Counter_Value"]))
def launches(pred):
    """[(bytes, name)] per launch: FETCH_SIZE x2 + WRITE_SIZE, both in units of 1 KB on gfx950 (the i-th launch of a name pairs its counters)"""
    out = []
    for name, d in per.items():
        if not pred(name):
            continue
        f, w = d.get("FETCH_SIZE", []), d.get("WRITE_SIZE", [])
        n = min(len(f), len(w))
        out += [((2.0 * f[i] + w[i]) * 1024.0, name) for i in range(n)]
    return out
def count(pred):
    return sum(min(len(d.get("FETCH_SIZE", [])), len(d.get("WRITE_SIZE", []))) for n, d in per.items() if pred(n))
def top_cluster(ls):
    """launches of the shape with the largest traffic (one instantiation can serve several shapes)"""
    if not ls:
        return []
    m = max(b for b, _ in ls)
    return [x for x in ls if x[0] > 0.6 * m]
def mean(ls):
    return sum(b for b, _ in ls) / len(ls) if ls else None
GEN = ("conv_", "bn_", "wgrad", "upsample2x", "sigmoid_bwd", "reflect_fold")
res, detail = {}, {}
fc2 = top_cluster(launches(lambda n: ("gemm_nt_kernel<" in n and ", 7u," in n) or "gemm8p_kernel<7u>" in n))
res["4"] = mean(fc2); detail["4"] = sorted({n[:60] for _, n in fc2})
# proj forward shares fc2's epilogue flags (7u): the launches BELOW fc2's cluster, above the patch embedding's (K = 192)
p7 = launches(lambda n: ("gemm_nt_kernel<" in n and ", 7u," in n) or "gemm8p_kernel<7u>" in n)
m7 = max((b for b, _ in p7), default=0.0)
pj = [x for x in p7 if 0.12 * m7 < x[0] <= 0.6 * m7]
res["9"] = mean(pj); detail["9"] = sorted({n[:60] for _, n in pj})
f1 = top_cluster(launches(lambda n: ("gemm_nt_kernel<" in n and ", 41u," in n) or "gemm8p_kernel<41u>" in n))
res["1"] = mean(f1); detail["1"] = sorted({n[:60] for _, n in f1})
qk = top_cluster(launches(lambda n: ("gemm_nt_kernel<" in n and ", 9u," in n) or "gemm8p_kernel<9u>" in n))
res["2"] = mean(qk); detail["2"] = sorted({n[:60] for _, n in qk})
bd = launches(lambda n: "gemm_nt_kernel<" in n and (", 72u," in n or ", 520u," in n))
res["11"] = mean(bd); detail["11"] = sorted({n[:60] for _, n in bd})
ln = launches(lambda n: "layernorm_fwd_kernel" in n or "layernorm_bwd_kernel" in n)
res["10"] = mean(ln); detail["10"] = sorted({n[:40] for _, n in ln})
dg = launches(lambda n: "gemm_nt_kernel<" in n and ", 4u," in n)
mx = max((b for b, _ in dg), default=0.0)
dg = [x for x in dg if x[0] > 0.25 * mx]          # (drops the patch-embedding dgrad, a 192-column GEMM with the same flags)
res["5"] = mean(dg); detail["5"] = sorted({n[:60] for _, n in dg})
af = launches(lambda n: "attn_fwd_kernel" in n or "attn_fwd8_kernel" in n or "attn_fwd_x32_kernel" in n)
res["3"] = mean(af); detail["3"] = sorted({n[:40] for _, n in af})
ab = launches(lambda n: "attn_bwd_kernel" in n or "attn_bwd_q_kernel" in n or "attn_bwd_kv_kernel" in n or "attn_bwd_x32_kernel" in n or "attn_bwd_q_x32_kernel" in n or "attn_bwd_kv_x32_kernel" in n)
# one call = one merged launch, or the dQ + dK/dV pair (counted by its dK/dV launch)
calls = count(lambda n: "attn_bwd_kernel" in n or "attn_bwd_x32_kernel" in n) + count(lambda n: "attn_bwd_kv_kernel" in n or "attn_bwd_kv_x32_kernel" in n)
res["6"] = sum(b for b, _ in ab) / calls if calls else None; detail["6"] = sorted({n[:40] for _, n in ab})
gen = launches(lambda n: any(k in n for k in GEN))
calls = 2 * count(lambda n: "conv_wgrad_batched_kernel<false>" in n)     # one such launch per backward call; as many forward calls
res["7"] = sum(b for b, _ in gen) / calls if calls else None; detail["7"] = f"{len(gen)} launches over {calls} calls"
ss = launches(lambda n: "selfsim_" in n)
calls = 2 * count(lambda n: "selfsim_dk_kernel" in n)
res["8"] = sum(b for b, _ in ss) / calls if calls else None; detail["8"] = sorted({n[:40] for _, n in ss})
res = {k: (None if v is None else round(v)) for k, v in res.items()}
res["_detail"] = detail
print(json.dumps(res, indent=1))
PY
