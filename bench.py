#!/usr/bin/env python3
"""Benchmark of the Splice per-pair optimisation step on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: either under a launcher -- python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
   ... bench.py --gpus N ... -- or bare: the script then starts its own N workers, one per GPU via HIP_VISIBLE_DEVICES)

Workload (config.workload): BASELINE.json configs[1] -- one 224x224 structure/appearance pair,
DINO ViT-B/8 (T = 785 tokens), bf16 ViT / fp32 generator, reference hyper-parameters
(conf/default/config.yaml), entire-image branch every 75th step, synthetic U[0,1) images and
seeded synthetic weights (no network for checkpoints).  One "step" = zero_grad, Model.forward,
LossG.forward, backward, Adam (train.py:56-80), nothing cached between steps.  Each rank
optimises its OWN pair on its own GPU (independent units, no data-path collective): weak scaling;
value = total steps of all ranks / max-over-ranks time.

Printed JSON (one line, rank 0) additionally carries
  roofline     : the live-timed kernel family with the largest share of the step (no family is excluded: at one 224 pair that is the
                 generator chain, reported per kernel -- its three longest kernels with their own FLOPs / bytes -- and as a roll-up): algorithmic
                 FLOPs of one call / the kernels' own begin-end time stamps; every other family under other_kernels
  north_star   : (attention forward + backward + key self-similarity FLOPs per step) / (their kernel time per step x the dense bf16 MFMA
                 peak) -- the figure BASELINE.json's north_star sets its 40 % target on
  cpu_baseline : the fp32 CPU oracle (a port of the reference-shaped loop: 6 ViT forwards + 3
                 backwards per step) timed on this host's cores on a bounded sample.

Hygiene: every SPLICE_* environment variable that changes what the library launches is echoed under config.env; the run is
REFUSED when one of them removes work or the default execution form (SPLICE_STEP_ABLATE, SPLICE_STEP_GRAPH=0,
SPLICE_STEP_OVERLAP=0, a library built with make DEV=1) unless --allow-dev-env is given, and then the line says so.

SPLICE_BENCH_STUB=<ms>: launcher self-test without a GPU (tests/test_bench_spawn_cpu.py) -- the engine is replaced by a fake
whose step sleeps <ms>; metric / data are renamed so that such a line can never be mistaken for a measurement.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import splice_amd  # noqa: E402,F401  (sets GPU_MAX_HW_QUEUES before the HIP runtime starts; see splice_amd/__init__.py)


def host_threads():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 64))


def library_env():
    """SPLICE_* variables the library reads (launch policies, debugging switches), minus bench.py's own plumbing."""
    own = {"SPLICE_BENCH_SPAWNED", "SPLICE_BENCH_STUB", "SPLICE_BENCH_STUB_GPUS"}
    return {k: v for k, v in sorted(os.environ.items()) if k.startswith("SPLICE_") and k not in own}


def dev_env_violations(dev_build):
    """Settings under which a timed step is not the product's step (the bench must refuse them)."""
    env, bad = library_env(), []
    for k in ("SPLICE_STEP_ABLATE", "SPLICE_STEP_SYNC"):   # on when non-zero
        if env.get(k, "0").strip() not in ("", "0"):
            bad.append(f"{k}={env[k]}")
    for k in ("SPLICE_STEP_GRAPH", "SPLICE_STEP_OVERLAP"):                         # the default execution form, off when 0
        if env.get(k, "1").strip() == "0":
            bad.append(f"{k}=0")
    if dev_build:
        bad.append("libsplice_hip.so built with make DEV=1 (timing-only switches compiled in)")
    return bad


class StubEngine:
    """Launcher self-test (SPLICE_BENCH_STUB): stands where synthetic_engine's result stands, does no GPU work."""
    def __init__(self, ms):
        self.ms = float(ms)
        self.step_idx = 0
        self.cfg = {"entire_A_every": 75}

    def step(self, A, B, E):
        time.sleep(self.ms * 1e-3)
        self.step_idx += 1

    def losses(self, *_):
        return {"loss": 0.0}


def pin_worker(rank, world, dev_index):
    """Host side of "within 5 % of linear" (SURVEY 8d): N workers share the node's cores -- each gets cores / N torch threads
    and, when the topology is readable, the CPUs of its GPU's NUMA node.  Best effort; what was done is reported per rank."""
    info = {"threads": None, "numa_node": None}
    try:
        import torch
        n = max(1, host_threads() // max(1, world))
        torch.set_num_threads(n)
        info["threads"] = n
        if world > 1 and torch.cuda.is_available():
            pr = torch.cuda.get_device_properties(dev_index)
            bdf = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read().strip())
            if node >= 0:
                cpus = set()
                for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
                    lo, _, hi = part.partition("-")
                    cpus.update(range(int(lo), int(hi or lo) + 1))
                cpus &= os.sched_getaffinity(0)
                if cpus:
                    os.sched_setaffinity(0, cpus)
                    info["numa_node"] = node
    except Exception:
        pass
    return info


def cpu_baseline(cfg, hw, seed, budget_s=25.0):
    """Reference-shaped fp32 loop (oracle/step.py) on the host cores; bounded sample."""
    import torch
    from oracle import dino_vit
    from oracle.step import SpliceOracle
    from splice_amd import synth
    threads = host_threads()
    torch.set_num_threads(threads)
    name = cfg["dino_model_name"]
    patch, dim, depth, heads = dino_vit.DINO_CONFIGS[name]
    m = dino_vit.VisionTransformer(patch, dim, depth, heads, img_size=cfg["dino_global_patch_size"]).eval()
    sd = synth.vit_params(seed, name, img_size=cfg["dino_global_patch_size"])
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    orc = SpliceOracle(m, {k: torch.from_numpy(v) for k, v in synth.generator_params(seed + 1, 0.02).items()}, cfg)
    A, B = synth.image_pair(seed, 0, hw[0], hw[1])
    A, B = torch.from_numpy(A)[None], torch.from_numpy(B)[None]
    orc.step(A, B, A)          # step 0 (entire branch, cls only): warm-up, untimed
    n, t0 = 0, time.time()
    while True:
        orc.step(A, B, A)      # ordinary steps (ssim + cls + id)
        n += 1
        el = time.time() - t0
        if el > budget_s or n >= 8:
            break
    return {"value": n / el, "unit": "steps/s", "cores": threads, "kind": "port",
            "sample": f"{n} ordinary steps (after 1 warm-up step) of the same 224x224 ViT-B/8 pair, fp32, "
                      f"reference-shaped 6-forward/3-backward loop, torch {torch.__version__} CPU"}


def spawn_workers(n, argv):
    """`python bench.py --gpus N` without a launcher: start N copies of this script, worker i pinned to GPU i through
    HIP_VISIBLE_DEVICES (one process per GPU), rendezvous on 127.0.0.1.  Worker 0 prints the JSON."""
    import socket
    import subprocess
    n_dev = visible_gpu_count()
    if n_dev < n:
        raise SystemExit(f"bench.py --gpus {n}: only {n_dev} GPU(s) visible on this node")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    parent_visible = os.environ.get("HIP_VISIBLE_DEVICES")
    ids = parent_visible.split(",") if parent_visible else [str(i) for i in range(n_dev)]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HIP_VISIBLE_DEVICES=ids[r], SPLICE_BENCH_SPAWNED="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        env.pop("CUDA_VISIBLE_DEVICES", None)
        env.pop("ROCR_VISIBLE_DEVICES", None)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rcs = [p.wait() for p in procs]
    if any(rcs):
        raise SystemExit(f"bench.py: worker exit codes {rcs}")


def visible_gpu_count():
    """GPUs this process could use, WITHOUT initialising the HIP runtime in the parent (the workers own the devices)."""
    import subprocess
    if os.environ.get("SPLICE_BENCH_STUB"):   # launcher self-test: pretend this many devices
        return int(os.environ.get("SPLICE_BENCH_STUB_GPUS", "0"))
    code = "import torch; print(torch.cuda.device_count() if torch.cuda.is_available() else 0)"
    try:
        return int(subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300).stdout.strip().splitlines()[-1])
    except Exception:
        return 0


# live-timed kernel families (splice_prof_begin): id -> (name, peak key, algorithmic FLOPs of the family's launches per step)
F32_MFMA_PEAK = 157.3     # TFLOP/s, exact-f32 MFMA = the f32 vector rate (MI355X_MICROARCH.md)
BF16_MFMA_PEAK = 2500.0   # TFLOP/s dense (also the rate of the non-scaled k = 32 fp8 MFMA)
FP8_MX_MFMA_PEAK = 5000.0 # TFLOP/s dense, block-scaled K = 128 e4m3 MFMA
HBM_PEAK = 8000.0         # GB/s


def kernel_families(T, D, heads, P, size, depth=12, fp8=False):
    """FLOPs are per host CALL of the family (one launch, or the launches of one call): a forward call covers 2 P passes
    (one of the two concurrent forward chains), a backward call P passes.  SURVEY.md 8d per-unit figures."""
    hidden = 4 * D
    g = (size / 224.0) ** 2
    return {
        # (tile / pipeline template arguments depend on the rows of a launch: <64,64,..,4> at one pair, <128,64,..,2> from 4 pairs on)
        # (--fp8: these three run on the block-scaled K = 128 e4m3 MFMA -- priced against ITS dense peak, 5 PFLOP/s)
        4: ("fc2 forward GEMM, BIAS|RESID|OUT_F32 epilogue (gemm_nt_kernel; gemm8p_kernel from ~150 row tiles of 256 on)" + (" [e4m3, scaled K=128 MFMA]" if fp8 else ""), "fp8mx" if fp8 else "bf16", 2 * P * 2.0 * T * hidden * D),
        1: ("fc1 forward GEMM, BIAS|GELU|OUT_BF epilogue (gemm_nt_kernel; gemm8p_kernel from 200 tiles of 256 x 256 on)" + (" [e4m3, scaled K=128 MFMA]" if fp8 else ""), "fp8mx" if fp8 else "bf16", 2 * P * 2.0 * T * hidden * D),
        2: ("qkv forward GEMM, BIAS|OUT_BF epilogue (gemm_nt_kernel; gemm8p_kernel from 200 tiles of 256 x 256 on)" + (" [e4m3, scaled K=128 MFMA]" if fp8 else ""), "fp8mx" if fp8 else "bf16", 2 * P * 2.0 * T * 3 * D * D),
        9: ("proj forward GEMM, BIAS|RESID|OUT_F32 epilogue (gemm_nt_kernel; gemm8p_kernel from ~150 row tiles of 256 on)", "bf16", 2 * P * 2.0 * T * D * D),
        3: ("attn_fwd8_kernel [e4m3 operands, k = 32 fp8 MFMA: bf16 rate]" if fp8 == "attention" else "attn_fwd_x32_kernel", "bf16", 2 * P * 4.0 * T * T * D),
        5: ("gemm_nt_kernel<OUT_F32> split-K dgrads (fc1^T and qkv^T, mean of both)", "bf16", P * 2.0 * T * D * (hidden + 3 * D) / 2),
        6: ("attn_bwd_x32_kernel (one launch, or dQ + dK/dV launches)", "bf16", P * 10.0 * T * T * D),
        # generator: one call = splice_gen_forward (2.262 GFLOP per 224^2 image) or splice_gen_backward (dgrad + wgrad = 2 x forward);
        # four calls per ordinary step (A plan, B plan; forward, backward) -> mean FLOPs per call = 1.5 x forward x P images
        7: ("generator chain: conv_igemm / BatchNorm / upsample / conv_wgrad kernels of one splice_gen_forward or _backward call", "f32",
            P * 1.5 * 2.2623e9 * g),
        # key self-similarity: S* (upper tiles, 0.47 of T^2 D x 2), S + loss (same), dK = W K (2 T^2 D); two calls per step (target | loss + dK)
        8: ("selfsim kernels (norms, S*, fused S / MSE / W, dK) per call", "bf16", P * (0.5 * 2.0 * T * T * D * 2 + 2.0 * T * T * D) / 2),
        # LayerNorm: HBM-bound; the work figure is algorithmic BYTES per call, mean of a forward call (2 P passes: fp32 row in, bf16 row out) and a
        # backward call (P passes: the split-K slabs of the dgrad in front of it -- three below 2401 rows per call, one from there on (gemm.h
        # GEMM_VSPLIT_ROWS: the sum is formed inside the GEMM) -- + x + incoming gradient in, fp32 gradient + its bf16 copy out = 14 + 4 x slabs B per element)
        10: ("layernorm_fwd_kernel / layernorm_bwd_kernel (mean of both)", "hbm", P * T * D * (2 * 6 + (14 + 4 * (3 if P * 800 < 2401 else 1))) / 2.0),
        11: ("gemm_nt_kernel bf16-output dgrads (fc2^T x GELU', proj^T + delta row dots; mean of both)", "bf16", P * 2.0 * T * D * (hidden + D) / 2),
    }


KERNEL_CLASSES = {"vit_gemms": (1, 2, 4, 9, 5, 11), "attention": (3, 6), "layernorm": (10,), "generator": (7,), "structure_loss": (8,)}


def hbm_copy_rate(dev):
    """This box's f32 copy rate (torch's float4 copy kernel over 1 GiB, GB/s moved = read + write): the boxes of the pool differ by a few per cent
    (-1.5 % on the headline between two rounds' boxes), so every line states what its box delivers."""
    import torch
    n = 1 << 28
    a = torch.empty(n, device=dev, dtype=torch.float32).fill_(1.0)
    b = torch.empty_like(a)
    for _ in range(2):
        b.copy_(a)
    torch.cuda.synchronize()
    best = 0.0
    for _ in range(3):
        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record()
        for _ in range(4):
            b.copy_(a)
        s1.record()
        torch.cuda.synchronize()
        best = max(best, 4 * 2.0 * n * 4 / (s0.elapsed_time(s1) * 1e-3) / 1e9)
    del a, b
    torch.cuda.empty_cache()
    return round(best, 1)


def time_steps(eng, A, B, K, W, barrier, E=None):
    E = A if E is None else E
    for _ in range(W):
        eng.step(A, B, E)
    barrier()
    t0 = time.perf_counter()
    for _ in range(K):
        eng.step(A, B, E)
    barrier()
    return time.perf_counter() - t0


MIN_TIMED_SECONDS = 1.0    # the driver's --steps 20 timed 0.07 s; a block of K steps is now repeated until >= 1 s is timed
MAX_BLOCKS = 400


def time_blocks(eng, A, B, K, W, barrier, rep, min_seconds=MIN_TIMED_SECONDS, max_blocks=MAX_BLOCKS, E=None):
    """W untimed warm-up steps, then BLOCKS of exactly K steps, each bracketed by barrier + synchronize on both sides and reduced to
    the max over ranks, repeated until at least ``min_seconds`` are timed.  Returns (local block times, max-over-ranks block times).
    Every rank sees the same reduced times, so every rank runs the same number of blocks."""
    E = A if E is None else E
    for _ in range(W):
        eng.step(A, B, E)
    local, reduced, total = [], [], 0.0
    while True:
        barrier()
        t0 = time.perf_counter()
        for _ in range(K):
            eng.step(A, B, E)
        barrier()
        dt = time.perf_counter() - t0
        dmax = rep.max_over_ranks(dt)
        local.append(dt)
        reduced.append(dmax)
        total += dmax
        if total >= min_seconds or len(reduced) >= max_blocks:
            return local, reduced


def median_block(reduced):
    """Index of the median block (lower median for an even count: a block that was actually run)."""
    order = sorted(range(len(reduced)), key=lambda i: reduced[i])
    return order[(len(order) - 1) // 2]


def timing_record(K, reduced, first_step, entire_every):
    mid = median_block(reduced)
    ms = [round(t / K * 1e3, 4) for t in reduced]
    ent = [sum(1 for st in range(first_step + b * K, first_step + (b + 1) * K) if entire_every and st % entire_every == 0) for b in range(len(reduced))]
    return mid, {"blocks": len(reduced), "steps_per_block": K, "timed_seconds": round(sum(reduced), 4), "median_block": mid,
                 "ms_per_step_median_block": ms[mid], "entire_image_steps_total": sum(ent),
                 "ms_per_step_by_block": ms if len(ms) <= 64 else ms[:64] + ["..."], "ms_per_step_min": min(ms), "ms_per_step_max": max(ms),
                 "entire_image_steps_by_block": ent if len(ent) <= 64 else ent[:64] + ["..."],
                 "rule": f"W warm-up steps, then blocks of exactly K steps (barrier + synchronize on both sides, max over ranks) until >= {MIN_TIMED_SECONDS} s "
                         "are timed; value / ms_per_step = ALL timed steps / ALL timed seconds (the periodic entire-image steps are in it at their true "
                         "frequency); the median block is kept as a robustness figure"}


def train_regime_leg(eng, A, B, steps=120):
    """the headline regime (fixed full crops: identity Resize, graph replay every step) is the BASELINE config, but
    ``train_model`` with the reference's default config draws a new crop size nearly every step (data/transforms.py:21: eager
    launches, bilinear Resize forward + adjoint), augments the structure image on the device (data/transforms.py:30-37) and
    runs a logging forward every ``log_images_freq`` = 10 steps.  The same engine is driven that way for ``steps`` steps
    (no PNG encode: that runs on a worker thread in train_model) and the rate is reported beside the headline."""
    import numpy as np
    import torch
    from splice_amd.train import DeviceDataFeed
    cfg = dict(eng.cfg, use_augmentations=True, global_A_crops_min_cover=0.95, global_B_crops_min_cover=0.95,
               global_A_crops_n_crops=1, global_B_crops_n_crops=1)
    np.random.seed(0)
    torch.manual_seed(0)
    feed = DeviceDataFeed(cfg, A.cpu(), B.cpu())
    feed.step = eng.step_idx          # keep the entire-image cadence aligned with the engine's step counter
    def run(n):
        for _ in range(n):
            inp = feed.next()
            log = (eng.step_idx + 2) % cfg["log_images_freq"] == 0
            if log:
                eng.generate(feed.get_A())
            eng.step(inp["A_global"], inp["B_global"], inp["A"][0] if "A" in inp else None)
            if log:
                eng.book_logged_forward()
    run(10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"steps_per_s": round(steps / dt, 2), "ms_per_step": round(dt / steps * 1e3, 4), "steps": steps,
            "regime": "random >= 95 % crops per step (new shapes: eager launches, bilinear Resize + adjoint), device augmentations, logging forward every 10 steps"}


def stub_main(args, ms, world):
    """Launcher self-test (no GPU): the same rendezvous, barrier, max-over-ranks timing and JSON assembly as the real run,
    around an engine whose step sleeps.  The line is labelled so it cannot pass for a measurement."""
    from splice_amd.dist import Replicas, aggregate_throughput
    rep = Replicas(backend="gloo", device=None)
    host = pin_worker(rep.local_rank, world, 0)
    eng = StubEngine(ms * (1.0 + 0.5 * rep.rank))   # rank r is slower: the max over ranks must show it
    local, reduced = time_blocks(eng, None, None, args.steps, args.warmup, rep.barrier, rep, min_seconds=float(os.environ.get("SPLICE_BENCH_STUB_MIN_S", "0.2")))
    mid, timing = timing_record(args.steps, reduced, args.warmup, 0)
    per_rank = rep.gather_floats(sum(local))
    elapsed = sum(reduced)
    k_total = args.steps * len(reduced)
    hip_vis = os.environ.get("HIP_VISIBLE_DEVICES", "")
    # every rank's device binding, gathered through the same process group (as floats: device ordinals)
    devs = rep.gather_floats(float(hip_vis.split(",")[0]) if hip_vis.split(",")[0].strip().isdigit() else -1.0)
    if rep.rank == 0:
        value = aggregate_throughput(k_total, world, elapsed)
        print(json.dumps({"metric": "stub_steps_per_sec", "value": round(value, 3), "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(elapsed / k_total * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "none", "data": "stub (launcher self-test: no GPU work, SPLICE_BENCH_STUB)",
                          "config": {"workload": f"STUB: sleep {ms} ms per step (+50 % per rank)", "gpus": world, "per_rank_steps_per_s": [round(k_total / t, 2) for t in per_rank],
                                     "per_rank_device": [int(d) for d in devs], "host": host, "env": library_env(), "timing": timing},
                          "roofline": None, "cpu_baseline": None}), flush=True)
    rep.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--size", type=int, default=224, help="pair height = width (configs[1]: 224)")
    ap.add_argument("--image", default="", help="HxW, e.g. 900x1200: the reference's DEFAULT workload shape (conf/default/config.yaml: 1200 x 900 images, A_resize -1) -- "
                                                "square global crops of the full image height (the 855 .. 900 range of data/transforms.py:21-23 at its upper end, fixed so that "
                                                "graphs replay), every crop resized to --size (224), the entire HxW image through Resize(224, max_size=480); one pair, no sweep")
    ap.add_argument("--model", default="dino_vitb8")
    ap.add_argument("--pairs", type=int, default=1, help="pairs optimised side by side per GPU in the timed region (1 = the reference's unit: the latency form of the metric)")
    ap.add_argument("--pairs-sweep", default="2,4,8,12,16,32", help="additional pairs-per-GPU settings timed briefly after the main region (throughput form: pairs/hr); '' = off")
    ap.add_argument("--fp8", nargs="?", const="gemm", default=None, choices=("gemm", "attention"),
                    help="fp8 operand path (own, looser tolerance table: an APPROXIMATE mode, tests/test_fp8_gpu.py).  '--fp8' = '--fp8 gemm': e4m3 QKV / fc1 / fc2 "
                         "projections + key self-similarity Gram on the fp8 MFMA (the fastest setting; the config key fp8: True); "
                         "'--fp8 attention': the attention forward on e4m3 operands too (BASELINE configs[4] as written)")
    ap.add_argument("--fp8-attention", choices=("on", "off"), default=None, help="(round-3 spelling, still accepted) with --fp8: 'on' = '--fp8 attention'")
    ap.add_argument("--scales", default="", help="BASELINE configs[4]: comma list of ViT input scales evaluated per step on the same crops (e.g. 224,320,448); "
                                                 "one fused step per scale + one Adam (MultiScaleEngine); disables the pairs sweep and the train-regime leg")
    ap.add_argument("--full-top-block", action="store_true", help="compute the whole top ViT block (default: behind its QKV projection only the [CLS] rows, "
                                                                  "all the losses read besides the keys)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train-regime", action="store_true", help="skip the train_model-shaped leg (random crops + augmentations + logging)")
    ap.add_argument("--prof-kernels", default="4,1,2,9,5,11,3,6,10,7,8", help="kernel families timed live for the roofline leg ('' = off; default: all): 4 fc2 fwd, 1 fc1 fwd, 2 qkv fwd, 9 proj fwd, "
                                                                  "5 split-K dgrads, 11 bf16-output dgrads, 3 attention fwd, 6 attention bwd, 10 LayerNorm, 7 generator chain, 8 key self-similarity")
    ap.add_argument("--allow-dev-env", action="store_true", help="run although a debugging / work-skipping SPLICE_* switch is set (the JSON line then carries config.dev_env)")
    args = ap.parse_args()
    fp8_mode = False if not args.fp8 else ("attention" if (args.fp8 == "attention" or args.fp8_attention == "on") else "gemm")   # splice_amd.vit.fp8_mode
    stub_ms = os.environ.get("SPLICE_BENCH_STUB")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_workers(args.gpus, sys.argv[1:])

    import torch
    from splice_amd.dist import Replicas, aggregate_throughput
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus} "
                         f"(or without a launcher: bench.py spawns its own workers)")
    if stub_ms:
        return stub_main(args, float(stub_ms), world)
    from splice_amd import _lib
    from splice_amd.engine import synthetic_engine
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    bad = dev_env_violations(bool(_lib.lib().splice_dev_switches()))
    if bad and not args.allow_dev_env:
        raise SystemExit("bench.py: refusing to time a step under " + "; ".join(bad) + " (debugging / work-skipping switches; --allow-dev-env overrides and marks the line)")
    # self-spawned workers see exactly one GPU each (HIP_VISIBLE_DEVICES); torchrun workers see all and pick LOCAL_RANK
    dev_index = 0 if os.environ.get("SPLICE_BENCH_SPAWNED") == "1" else local_rank
    if dev_index >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {local_rank} has no GPU (visible devices: {torch.cuda.device_count()})")
    torch.cuda.set_device(dev_index)
    dev = f"cuda:{dev_index}"
    host = pin_worker(local_rank, world, dev_index)
    # The replicas exchange NO data (independent pairs): the process group exists only for the start/stop barrier and the
    # max-over-ranks of one float, so it runs over gloo on the host -- an RCCL communicator would add an xGMI bootstrap that
    # can only hurt (north_star: "no RCCL required").
    rep = Replicas(backend="gloo", device=None)
    rank = rep.rank

    cfg = dict(dino_model_name=args.model, dino_global_patch_size=args.size)
    hw = (args.size, args.size)
    P = max(1, args.pairs)
    scales = [int(x) for x in args.scales.split(",") if x.strip()]
    if scales:
        from splice_amd import synth
        from splice_amd.engine import MultiScaleEngine
        P = 1
        args.pairs_sweep, args.no_train_regime, args.no_cpu_baseline = "", True, True
        Ai, Bi = synth.image_pair(1234, rep.pair_id(), hw[0], hw[1])
        eng = MultiScaleEngine(cfg, synth.vit_params(1234, args.model, img_size=224), synth.generator_params(1235 + rep.pair_id(), 0.02), hw, hw,
                               scales=scales, device=dev, fp8=fp8_mode)
        A, B = torch.from_numpy(Ai).to(dev), torch.from_numpy(Bi).to(dev)
    elif args.image:
        ih, iw = (int(x) for x in args.image.lower().split("x"))
        side = min(ih, iw)
        P = 1
        args.pairs_sweep, args.no_cpu_baseline = "", True
        eng, A, B, E_img = synthetic_engine(cfg, pair_id=rep.pair_id(), hw=(ih, iw), seed=1234, device=dev, fp8=fp8_mode, crop_hw=(side, side))
        from splice_amd import synth as _synth
        B_full = torch.from_numpy(_synth.image_pair(1234, rep.pair_id(), ih, iw)[1]).to(dev)   # (the train_model regime draws its random 95 .. 100 % crops from the full images)
        hw = (ih, iw)
    else:
        eng, A, B = synthetic_engine(cfg, pair_id=rep.pair_id() * P, hw=hw, seed=1234, device=dev, pairs=P, fp8=fp8_mode, top_cls_only=not args.full_top_block)
    E_in = E_img if args.image else None   # the entire image (the crops themselves at the square BASELINE configs)
    gen_side = min(hw) if args.image else args.size   # side of the square generator crops (the roofline leg's FLOP count)
    K, W = args.steps, args.warmup

    def barrier():
        torch.cuda.synchronize()
        rep.barrier()
        torch.cuda.synchronize()

    local_blocks, reduced_blocks = time_blocks(eng, A, B, K, W, barrier, rep, E=E_in)
    mid, timing = timing_record(K, reduced_blocks, W, eng.cfg["entire_A_every"] if not scales else eng.engines[0].cfg["entire_A_every"])
    n_blocks = len(reduced_blocks)
    elapsed = sum(local_blocks)          # this rank's timed seconds over all blocks
    losses = eng.losses() if (P == 1 or scales) else eng.losses(0)
    # roofline leg: the timed region replays captured hipGraphs (event records cannot be threaded through a
    # replay), so the SAME steps continue for short instrumented stretches with every launch of one kernel family
    # bracketed by HIP events on its own stream (eager launches; the kernels themselves are identical).
    prof = {}
    fam_ids = [int(x) for x in args.prof_kernels.split(",") if x.strip()] if rank == 0 else []
    for fam in fam_ids:
        ms, calls, kernels = C.c_float(0), C.c_int(0), C.c_int(0)
        _lib.check(_lib.lib().splice_prof_begin(fam))
        nprof = min(K, 20)
        for _ in range(nprof):
            eng.step(A, B, A if E_in is None else E_in)
        torch.cuda.synchronize()
        detail = C.create_string_buffer(16384)
        _lib.check(_lib.lib().splice_prof_end_detail(C.byref(ms), C.byref(calls), C.byref(kernels), detail, len(detail)))
        if calls.value:
            n_ent = sum(1 for st_ in range(eng.step_idx - nprof + 1, eng.step_idx + 1) if st_ % eng.cfg["entire_A_every"] == 0)
            per_kernel = []
            for line in detail.value.decode(errors="replace").splitlines():
                nm, n, tms, fl, by = line.split("\t")
                per_kernel.append({"kernel": nm, "launches": int(n), "ms": float(tms), "flops": float(fl), "bytes": float(by)})
            prof[fam] = (ms.value, calls.value, kernels.value, nprof, n_ent, per_kernel)
    per_rank_elapsed = rep.gather_floats(elapsed)
    elapsed = sum(reduced_blocks)        # max over ranks, block by block
    K_total = K * n_blocks
    T = eng.ctx_g.T if not scales else [e.ctx_g.T for e in eng.engines]
    D = eng.vit.dim
    n_entire = timing["entire_image_steps_total"]
    # ---- throughput form of the metric: P pairs per GPU through the shared ViT (same barrier-bracketed timing, fewer steps)
    sweep = {P: K_total / elapsed}
    sweep_ids = [int(x) for x in args.pairs_sweep.split(",") if x.strip()] if (world == 1 and P == 1 and not scales) else []
    vit = eng.vit
    for Ps in sweep_ids:
        try:
            e2, A2, B2 = synthetic_engine(cfg, pair_id=0, hw=hw, seed=1234, device=dev, pairs=Ps, vit_engine=vit, fp8=fp8_mode, top_cls_only=not args.full_top_block)
            k2 = max(20, K // 4) if Ps <= 8 else 12   # (the big batches: ~0.4 / 0.8 s of steps at 16 / 32 pairs)
            sweep[Ps] = k2 / time_steps(e2, A2, B2, k2, max(5, W // 2) if Ps <= 8 else 4, barrier)
            del e2, A2, B2
            torch.cuda.empty_cache()
        except Exception as e:   # the sweep must never take the headline number down
            sweep[Ps] = None
            print(f"[bench] pairs={Ps} sweep leg failed: {e}", file=sys.stderr)
    train_leg = None
    if world == 1 and P == 1 and args.size >= 64 and not args.no_train_regime:
        try:
            train_leg = train_regime_leg(eng, E_img, B_full) if args.image else train_regime_leg(eng, A, B)
        except Exception as e:
            train_leg = {"steps_per_s": None, "regime": f"failed: {e}"}
    if rank != 0:
        rep.close()
        return

    if scales:   # every scale makes the same host calls per step: a family's mean FLOPs per call = the mean over the scales
        per = [kernel_families(t, D, eng.vit.heads, P, args.size, fp8=fp8_mode) for t in T]
        fams = {k: (per[0][k][0] + f" (mean over the ViT input scales {scales})", per[0][k][1], sum(f[k][2] for f in per) / len(per)) for k in per[0]}
        fams[7] = per[0][7]   # (the generator works at the crop size at every scale)
    else:
        fams = kernel_families(T, D, eng.vit.heads, P, gen_side, fp8=fp8_mode)
    roofs = []
    traffic_file = os.path.join(ROOT, "profiles", "roofline_traffic.json")
    static_traffic = {}
    if os.path.exists(traffic_file) and args.model == "dino_vitb8" and not scales and not args.fp8:
        try:   # keys: "P<pairs>" at 224 x 224, "S<size>P<pairs>" otherwise, "IMG<h>x<w>" for --image; values: family id -> bytes per call
            tkey = f"IMG{hw[0]}x{hw[1]}" if args.image else (f"P{P}" if args.size == 224 else f"S{args.size}P{P}")
            static_traffic = json.load(open(traffic_file)).get(tkey, {})
        except Exception:
            static_traffic = {}
    for fam, (tot_ms, calls, kernels, steps, n_ent, per_kernel) in prof.items():
        kname, cls, flops = fams[fam]
        avg_ms = max(tot_ms / calls, 1e-6)
        hbm_bound = cls == "hbm"   # (work = algorithmic bytes per call)
        ach = flops / (avg_ms * 1e-3) / (1e9 if hbm_bound else 1e12)
        peak = HBM_PEAK if hbm_bound else {"bf16": BF16_MFMA_PEAK, "fp8mx": FP8_MX_MFMA_PEAK}.get(cls, F32_MFMA_PEAK)
        traffic = static_traffic.get(str(fam))
        r = {"bound": "hbm" if hbm_bound else "mfma", "family": fam, "kernel": kname, "achieved": round(ach, 2), "peak": peak, "unit": "GB/s" if hbm_bound else "TFLOP/s",
             "frac": round(ach / peak, 4), "traffic": traffic,
             "traffic_source": None if traffic is None else "static: profiles/roofline_traffic.json, bytes per call of this family (FETCH_SIZE x2 + WRITE_SIZE from separate --pmc passes of tools/pmc_families.sh on this workload and build; not re-measured by this command)",
             "avg_launch_us": round(avg_ms * 1e3, 2), "calls_per_step": round(calls / steps, 1), "kernels_per_step": round(kernels / steps, 1),
             "share_of_step_ms": round(tot_ms / steps, 4)}
        if len(per_kernel) > 1:   # a chain of kernels: its three longest, each with its own launch count, duration and (where the launcher notes it) work
            top = []
            for k in per_kernel[:3]:
                e = {"kernel": k["kernel"], "launches_per_step": round(k["launches"] / steps, 1), "avg_launch_us": round(k["ms"] / k["launches"] * 1e3, 2),
                     "share_of_step_ms": round(k["ms"] / steps, 4)}
                if k["flops"] > 0:
                    tf = k["flops"] / (k["ms"] * 1e-3) / 1e12
                    gb = k["bytes"] / (k["ms"] * 1e-3) / 1e9
                    e.update({"algorithmic_gflop_per_launch": round(k["flops"] / k["launches"] / 1e9, 4), "achieved_tflops": round(tf, 2), "frac_of_mfma_peak": round(tf / peak, 4),
                              "algorithmic_mb_per_launch": round(k["bytes"] / k["launches"] / 1e6, 3), "achieved_gb_s": round(gb, 1), "frac_of_hbm_peak": round(gb / HBM_PEAK, 4)})
                top.append(e)
            r["kernels"] = top
        if n_ent:
            r["note_entire"] = f"{n_ent} of the {steps} instrumented steps ran the entire-image branch (its launches are in the averages)"
        if fam == 7:   # also against the HBM roofline: SURVEY 8d floor of 11.54 M fp32 activation elements per image forward, backward 2x
            g = (gen_side / 224.0) ** 2
            bytes_call = P * 1.5 * 11.54e6 * 4 * g
            r["hbm"] = {"bound": "hbm", "algorithmic_bytes_per_call": round(bytes_call), "achieved": round(bytes_call / (avg_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK,
                        "unit": "GB/s", "frac": round(bytes_call / (avg_ms * 1e-3) / 1e9 / HBM_PEAK, 4)}
        roofs.append(r)
    roofs.sort(key=lambda r: -r["share_of_step_ms"])
    roof = None
    if roofs:
        # dominant = the single KERNEL family with the largest total time per step (the generator entry is a chain of ~70 kernels
        # per call and is listed with the others; `furthest_below_roofline` names the family with the lowest fraction)
        singles = roofs   # (no family is excluded: the largest share is the dominant one)
        roof = dict(singles[0])
        roof["note"] = ("dominant = the live-timed kernel family with the largest total time per step; achieved = algorithmic FLOPs of one "
                        "call / mean kernel duration of its launches; the duration is the kernel's own begin / end time stamp pair "
                        "(hipExtLaunchKernelGGL start / stop events on the launch stream, nothing subtracted), taken over an instrumented "
                        "continuation of the timed steps (the timed region itself replays hipGraphs)")
        roof["other_kernels"] = [r for r in roofs if r is not singles[0]]
        worst = min(roofs, key=lambda r: r["frac"])
        roof["furthest_below_roofline"] = {"kernel": worst["kernel"], "frac": worst["frac"], "share_of_step_ms": worst["share_of_step_ms"]}
    by_class = None
    if prof:   # class roll-up: every live-timed family in one of five classes -- which CLASS dominates does not depend on how the families are cut
        by_class = {}
        for cname, ids in KERNEL_CLASSES.items():
            got = [f for f in ids if f in prof]
            if not got:
                continue
            ms_step = sum(prof[f][0] / prof[f][3] for f in got)
            launches = sum(prof[f][2] / prof[f][3] for f in got)
            work = sum(fams[f][2] * prof[f][1] / prof[f][3] for f in got)   # algorithmic FLOPs (bytes for LayerNorm) per step
            cls0 = fams[got[0]][1]
            if cls0 == "hbm":
                rate, pk, unit = work / (ms_step * 1e-3) / 1e9, HBM_PEAK, "GB/s"
            else:
                rate, pk, unit = work / (ms_step * 1e-3) / 1e12, {"bf16": BF16_MFMA_PEAK, "fp8mx": FP8_MX_MFMA_PEAK}.get(cls0, F32_MFMA_PEAK), "TFLOP/s"
            by_class[cname] = {"families": got, "families_missing": [f for f in ids if f not in prof], "kernel_ms_per_step": round(ms_step, 4), "launches_per_step": round(launches, 1),
                               ("algorithmic_gb_per_step" if cls0 == "hbm" else "algorithmic_gflop_per_step"): round(work / 1e9, 3), "achieved": round(rate, 2), "peak": pk, "unit": unit,
                               "frac": round(rate / pk, 4)}
        tot_ms = sum(v["kernel_ms_per_step"] for v in by_class.values())
        for v in by_class.values():
            v["share_of_timed_kernel_ms"] = round(v["kernel_ms_per_step"] / tot_ms, 4)
        if roof is not None:
            roof["by_class"] = by_class
            roof["by_class_note"] = ("serialised kernel time of the live-timed families per step (the step itself overlaps two chains: ms_per_step is smaller than the sum); not live-timed: "
                                     "patch embedding, the [CLS]-row tail of the top block, resize / staging / Adam (< 5 % of kernel time, profiles/r06_kernel_stats_p1.csv)")
    north = None
    if all(f in prof for f in (3, 6, 8)):   # BASELINE north_star: MFMA utilisation of the attention + key self-similarity kernels
        fl = sum(fams[f][2] * prof[f][1] / prof[f][3] for f in (3, 6, 8))            # algorithmic FLOPs per step
        ms_ = sum(prof[f][0] / prof[f][3] for f in (3, 6, 8))                         # kernel ms per step
        north = {"what": "(attention forward + backward + key self-similarity algorithmic FLOPs per step) / (their kernel time per step x dense bf16 MFMA peak)",
                 "gflop_per_step": round(fl / 1e9, 2), "kernel_ms_per_step": round(ms_, 4), "achieved_tflops": round(fl / (ms_ * 1e-3) / 1e12, 1),
                 "peak_tflops": BF16_MFMA_PEAK, "frac": round(fl / (ms_ * 1e-3) / 1e12 / BF16_MFMA_PEAK, 4), "target": 0.40,
                 "by_family": {fams[f][0].split(" ")[0]: round(fams[f][2] / (prof[f][0] / prof[f][1] * 1e-3) / 1e12 / BF16_MFMA_PEAK, 4) for f in (3, 6, 8)}}
    try:
        box_copy = hbm_copy_rate(dev)
    except Exception:
        box_copy = None
    cpu = None
    if not args.no_cpu_baseline and world == 1:
        try:
            cpu = cpu_baseline(dict(eng.cfg), hw, 1234)
        except Exception as e:  # the baseline must never take the product number down
            cpu = {"value": None, "unit": "steps/s", "cores": host_threads(), "kind": "port", "sample": f"failed: {e}"}
    value = aggregate_throughput(K_total, world, elapsed)
    best_P = max((k for k, v in sweep.items() if v), key=lambda k: sweep[k] * k)
    out = {
        "metric": "opt_steps_per_sec", "value": round(value, 3), "unit": "steps/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": round(elapsed / K_total * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": ("fp8-approximate(e4m3 qkv/fc1/fc2 + attention fwd + selfsim Gram)/bf16" if fp8_mode == "attention" else "fp8-approximate(e4m3 qkv/fc1/fc2 + selfsim Gram)/bf16") if args.fp8 else "bf16", "data": "synthetic",
        "config": {"workload": (f"Splice pair {hw[0]}x{hw[1]} (the reference's default image shape), global crops {gen_side}x{gen_side} resized to {args.size}, entire image through Resize({args.size}, max_size=480), "
                                if args.image else f"Splice pair {hw[0]}x{hw[1]}, ") + f"{args.model} (T={T}), {P} pair(s) per GPU per step, "
                               f"{n_entire} of the {K_total} timed steps ({n_blocks} blocks of {K}) include the entire-image branch"
                               + (f"; loss evaluated at the ViT input scales {scales} every step (configs[4])" if scales else "")
                               + ("; APPROXIMATE fp8 operand mode (per-step gradient 1e-1 off the fp32 oracle, own tolerance table: tests/test_fp8_gpu.py, DESIGN.md section 5)" if args.fp8 else ""),
                   "gpus": world, "pairs_per_gpu": P, "pair_steps_per_s": round(value * P, 3),
                   "pairs_per_hour_at_2000_steps": round(value * P * 3600 / 2000, 2),
                   "per_rank_steps_per_s": [round(K_total / t, 2) for t in per_rank_elapsed],
                   "best_pairs_per_gpu": {"pairs": best_P, "pair_steps_per_s": round(sweep[best_P] * best_P, 2), "pairs_per_hour_at_2000_steps": round(sweep[best_P] * best_P * 3600 / 2000, 1)},
                   "throughput_by_pairs_per_gpu": {str(k): (None if v is None else {"steps_per_s": round(v, 2), "pair_steps_per_s": round(v * k, 2),
                                                                                    "pairs_per_hour_at_2000_steps": round(v * k * 3600 / 2000, 1)})
                                                   for k, v in sorted(sweep.items())},
                   "train_model_regime": train_leg,
                   "generator_dtype": "f32", "last_loss": round(losses["loss"], 5), "box_hbm_f32_copy_gb_s": box_copy,
                   "env": library_env(), "host": host, "timing": timing},
        "roofline": roof, "north_star": north, "cpu_baseline": cpu,
    }
    if bad:
        out["config"]["dev_env"] = bad   # --allow-dev-env: NOT a measurement of the product's step
    print(json.dumps(out), flush=True)
    rep.close()


if __name__ == "__main__":
    main()
