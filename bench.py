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
  roofline     : the dominant kernel (fc2 bf16 MFMA GEMM of the ViT forward, the largest single share of the step): algorithmic
                 FLOPs per launch / live HIP-event duration of its launches inside the timed region
  cpu_baseline : the fp32 CPU oracle (a port of the reference-shaped loop: 6 ViT forwards + 3
                 backwards per step) timed on this host's cores on a bounded sample.
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
    code = "import torch; print(torch.cuda.device_count() if torch.cuda.is_available() else 0)"
    try:
        return int(subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300).stdout.strip().splitlines()[-1])
    except Exception:
        return 0


# live-timed kernel families (splice_prof_begin): name, launches cover `passes(P)` ViT passes, FLOPs per pass
def kernel_families(T, D, heads, P):
    hidden = 4 * D
    return {
        # (tile / pipeline template arguments depend on the rows of a launch: <64,64,..,4> at one pair, <128,64,..,2> from 4 pairs on)
        4: ("gemm_nt_kernel<BIAS|RESID|OUT_F32> fc2 forward", 2 * P, 2.0 * T * hidden * D),
        1: ("gemm_nt_kernel<BIAS|GELU|OUT_BF> fc1 forward", 2 * P, 2.0 * T * hidden * D),
        2: ("gemm_nt_kernel<BIAS|OUT_BF|OUT_T> qkv forward", 2 * P, 2.0 * T * 3 * D * D),
        3: ("attn_fwd_kernel", 2 * P, 4.0 * T * T * D),
        5: ("gemm_nt_kernel<OUT_F32> split-K dgrads (fc1^T and qkv^T, mean of both)", P, 2.0 * T * D * (hidden + 3 * D) / 2),
        6: ("attn_bwd_kernel (merged, or dQ + dK/dV launches)", P, 10.0 * T * T * D),
    }


def time_steps(eng, A, B, K, W, barrier):
    for _ in range(W):
        eng.step(A, B, A)
    barrier()
    t0 = time.perf_counter()
    for _ in range(K):
        eng.step(A, B, A)
    barrier()
    return time.perf_counter() - t0


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--size", type=int, default=224, help="pair height = width (configs[1]: 224)")
    ap.add_argument("--model", default="dino_vitb8")
    ap.add_argument("--pairs", type=int, default=1, help="pairs optimised side by side per GPU in the timed region (1 = the reference's unit: the latency form of the metric)")
    ap.add_argument("--pairs-sweep", default="2,4,8", help="additional pairs-per-GPU settings timed briefly after the main region (throughput form: pairs/hr); '' = off")
    ap.add_argument("--fp8", action="store_true", help="BASELINE configs[4] operand path: QKV projections + key self-similarity Gram on the fp8 MFMA (own tolerance table)")
    ap.add_argument("--scales", default="", help="BASELINE configs[4]: comma list of ViT input scales evaluated per step on the same crops (e.g. 224,320,448); "
                                                 "one fused step per scale + one Adam (MultiScaleEngine); disables the sweep / roofline / train-regime legs")
    ap.add_argument("--full-top-block", action="store_true", help="compute the whole top ViT block (default: behind its QKV projection only the [CLS] rows, "
                                                                  "all the losses read besides the keys)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train-regime", action="store_true", help="skip the train_model-shaped leg (random crops + augmentations + logging)")
    ap.add_argument("--prof-kernels", default="4,5,3,6", help="kernel families timed live with HIP events for the roofline leg ('' = off): 4 fc2 fwd, 1 fc1 fwd, 2 qkv fwd, 3 attention fwd, 5 split-K dgrads, 6 attention bwd")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_workers(args.gpus, sys.argv[1:])

    import torch
    from splice_amd import _lib
    from splice_amd.engine import synthetic_engine

    from splice_amd.dist import Replicas, aggregate_throughput
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus} "
                         f"(or without a launcher: bench.py spawns its own workers)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    # self-spawned workers see exactly one GPU each (HIP_VISIBLE_DEVICES); torchrun workers see all and pick LOCAL_RANK
    dev_index = 0 if os.environ.get("SPLICE_BENCH_SPAWNED") == "1" else local_rank
    if dev_index >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {local_rank} has no GPU (visible devices: {torch.cuda.device_count()})")
    torch.cuda.set_device(dev_index)
    dev = f"cuda:{dev_index}"
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
        args.pairs_sweep, args.prof_kernels, args.no_train_regime, args.no_cpu_baseline = "", "", True, True
        Ai, Bi = synth.image_pair(1234, rep.pair_id(), hw[0], hw[1])
        eng = MultiScaleEngine(cfg, synth.vit_params(1234, args.model, img_size=224), synth.generator_params(1235 + rep.pair_id(), 0.02), hw, hw,
                               scales=scales, device=dev, fp8=args.fp8)
        A, B = torch.from_numpy(Ai).to(dev), torch.from_numpy(Bi).to(dev)
    else:
        eng, A, B = synthetic_engine(cfg, pair_id=rep.pair_id() * P, hw=hw, seed=1234, device=dev, pairs=P, fp8=args.fp8, top_cls_only=not args.full_top_block)
    K, W = args.steps, args.warmup

    def barrier():
        torch.cuda.synchronize()
        rep.barrier()
        torch.cuda.synchronize()

    elapsed = time_steps(eng, A, B, K, W, barrier)
    losses = eng.losses() if (P == 1 or scales) else eng.losses(0)
    # roofline leg: the timed region replays captured hipGraphs (event records cannot be threaded through a
    # replay), so the SAME steps continue for short instrumented stretches with every launch of one kernel family
    # bracketed by HIP events on its own stream (eager launches; the kernels themselves are identical).
    prof = {}
    ev_overhead_ms = 0.0
    fam_ids = [int(x) for x in args.prof_kernels.split(",") if x.strip()] if rank == 0 else []
    if fam_ids:
        # an event pair costs time by itself: calibrate on empty pairs (same stream) and subtract
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(200)]
        for a_, b_ in evs:
            a_.record(); b_.record()
        torch.cuda.synchronize()
        ev_overhead_ms = sorted(a_.elapsed_time(b_) for a_, b_ in evs)[len(evs) // 2]
        for fam in fam_ids:
            ms, n = C.c_float(0), C.c_int(0)
            _lib.check(_lib.lib().splice_prof_begin(fam))
            for _ in range(min(K, 20)):
                eng.step(A, B, A)
            torch.cuda.synchronize()
            _lib.check(_lib.lib().splice_prof_end(C.byref(ms), C.byref(n)))
            if n.value:
                prof[fam] = (ms.value, n.value, min(K, 20))
    per_rank_elapsed = rep.gather_floats(elapsed)
    elapsed = rep.max_over_ranks(elapsed)
    T = eng.ctx_g.T if not scales else [e.ctx_g.T for e in eng.engines]
    D = eng.vit.dim
    n_entire = sum(1 for s in range(W, W + K) if s % eng.cfg["entire_A_every"] == 0)
    # ---- throughput form of the metric: P pairs per GPU through the shared ViT (same barrier-bracketed timing, fewer steps)
    sweep = {P: K / elapsed}
    sweep_ids = [int(x) for x in args.pairs_sweep.split(",") if x.strip()] if (world == 1 and P == 1 and not scales) else []
    vit = eng.vit
    for Ps in sweep_ids:
        try:
            e2, A2, B2 = synthetic_engine(cfg, pair_id=0, hw=hw, seed=1234, device=dev, pairs=Ps, vit_engine=vit, fp8=args.fp8, top_cls_only=not args.full_top_block)
            k2 = max(20, K // 4)
            sweep[Ps] = k2 / time_steps(e2, A2, B2, k2, max(5, W // 2), barrier)
            del e2, A2, B2
            torch.cuda.empty_cache()
        except Exception as e:   # the sweep must never take the headline number down
            sweep[Ps] = None
            print(f"[bench] pairs={Ps} sweep leg failed: {e}", file=sys.stderr)
    train_leg = None
    if world == 1 and P == 1 and args.size >= 64 and not args.no_train_regime:
        try:
            train_leg = train_regime_leg(eng, A, B)
        except Exception as e:
            train_leg = {"steps_per_s": None, "regime": f"failed: {e}"}
    if rank != 0:
        rep.close()
        return

    fams = kernel_families(T if not scales else T[0], D, eng.vit.heads, P)
    roofs = []
    traffic_file = os.path.join(ROOT, "profiles", "roofline_traffic.json")
    static_traffic = {}
    if os.path.exists(traffic_file) and args.size == 224 and args.model == "dino_vitb8" and not scales and not args.fp8:
        try:
            static_traffic = json.load(open(traffic_file)).get(f"P{P}", {})
        except Exception:
            static_traffic = {}
    for fam, (tot_ms, n, steps) in prof.items():
        kname, passes, flops_per_pass = fams[fam]
        flops = flops_per_pass * passes
        avg_ms = max(tot_ms / n - ev_overhead_ms, 1e-6)
        ach = flops / (avg_ms * 1e-3) / 1e12
        traffic = static_traffic.get(str(fam))
        roofs.append({"bound": "mfma", "kernel": kname, "achieved": round(ach, 2), "peak": 2500.0, "unit": "TFLOP/s",
                      "frac": round(ach / 2500.0, 4), "traffic": traffic,
                      "traffic_source": None if traffic is None else "static: profiles/roofline_traffic.json (FETCH_SIZE x2 + WRITE_SIZE from separate --pmc passes of tools/pmc_traffic.sh on this workload, round 2; not re-measured by this command)",
                      "avg_launch_us": round(avg_ms * 1e3, 2), "launches_per_step": round(n / steps, 1),
                      "share_of_step_ms": round((tot_ms - n * ev_overhead_ms) / steps, 4), "event_pair_overhead_us": round(ev_overhead_ms * 1e3, 2)})
    roofs.sort(key=lambda r: -r["share_of_step_ms"])
    roof = None
    if roofs:
        roof = dict(roofs[0])
        roof["note"] = ("dominant = the live-timed kernel family with the largest total time per step; achieved = algorithmic FLOPs of one "
                        "launch / mean HIP-event duration of its launches, measured on the launch stream over an instrumented continuation "
                        "of the timed steps (the timed region itself replays hipGraphs), minus the median cost of an empty event pair")
        roof["other_kernels"] = roofs[1:]
    cpu = None
    if not args.no_cpu_baseline and world == 1:
        try:
            cpu = cpu_baseline(dict(eng.cfg), hw, 1234)
        except Exception as e:  # the baseline must never take the product number down
            cpu = {"value": None, "unit": "steps/s", "cores": host_threads(), "kind": "port", "sample": f"failed: {e}"}
    value = aggregate_throughput(K, world, elapsed)
    out = {
        "metric": "opt_steps_per_sec", "value": round(value, 3), "unit": "steps/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": round(elapsed / K * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "fp8(e4m3 qkv+selfsim)/bf16" if args.fp8 else "bf16", "data": "synthetic",
        "config": {"workload": f"Splice pair {hw[0]}x{hw[1]}, {args.model} (T={T}), {P} pair(s) per GPU per step, "
                               f"{n_entire} of {K} timed steps include the entire-image branch"
                               + (f"; loss evaluated at the ViT input scales {scales} every step (configs[4])" if scales else ""),
                   "gpus": world, "pairs_per_gpu": P, "pair_steps_per_s": round(value * P, 3),
                   "pairs_per_hour_at_2000_steps": round(value * P * 3600 / 2000, 2),
                   "per_rank_steps_per_s": [round(K / t, 2) for t in per_rank_elapsed],
                   "throughput_by_pairs_per_gpu": {str(k): (None if v is None else {"steps_per_s": round(v, 2), "pair_steps_per_s": round(v * k, 2),
                                                                                    "pairs_per_hour_at_2000_steps": round(v * k * 3600 / 2000, 1)})
                                                   for k, v in sorted(sweep.items())},
                   "train_model_regime": train_leg,
                   "generator_dtype": "f32", "last_loss": round(losses["loss"], 5)},
        "roofline": roof, "cpu_baseline": cpu,
    }
    print(json.dumps(out), flush=True)
    rep.close()


if __name__ == "__main__":
    main()
