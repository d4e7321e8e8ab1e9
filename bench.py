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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--size", type=int, default=224, help="pair height = width (configs[1]: 224)")
    ap.add_argument("--model", default="dino_vitb8")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--prof-kernel", type=int, default=4, help="4 fc2 GEMM (largest share of the step), 1 fc1 GEMM, 2 qkv GEMM, 3 attention fwd, 0 off")
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
    eng, A, B = synthetic_engine(cfg, pair_id=rep.pair_id(), hw=hw, seed=1234, device=dev)
    K, W = args.steps, args.warmup

    def barrier():
        torch.cuda.synchronize()
        rep.barrier()
        torch.cuda.synchronize()

    for _ in range(W):
        eng.step(A, B, A)
    barrier()
    t0 = time.perf_counter()
    for _ in range(K):
        eng.step(A, B, A)
    barrier()
    elapsed = time.perf_counter() - t0
    losses = eng.losses()
    # roofline leg: the timed region replays captured hipGraphs (event records cannot be threaded through a
    # replay), so the SAME steps continue for a short instrumented stretch with every launch of the chosen
    # kernel bracketed by HIP events on its own stream (eager launches; the kernel itself is identical).
    prof_ms, prof_n = C.c_float(0), C.c_int(0)
    if args.prof_kernel and rank == 0:
        _lib.check(_lib.lib().splice_prof_begin(args.prof_kernel))
        for _ in range(min(K, 30)):
            eng.step(A, B, A)
        torch.cuda.synchronize()
        _lib.check(_lib.lib().splice_prof_end(C.byref(prof_ms), C.byref(prof_n)))
        # an event pair costs time by itself: calibrate on empty pairs (same stream) and subtract
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(200)]
        for a_, b_ in evs:
            a_.record(); b_.record()
        torch.cuda.synchronize()
        ev_overhead_ms = sorted(a_.elapsed_time(b_) for a_, b_ in evs)[len(evs) // 2]
    else:
        ev_overhead_ms = 0.0
    per_rank_elapsed = rep.gather_floats(elapsed)
    elapsed = rep.max_over_ranks(elapsed)
    if rank != 0:
        rep.close()
        return

    T = eng.ctx_g.T
    D, hidden = eng.vit.dim, 4 * eng.vit.dim
    n_entire = sum(1 for s in range(W, W + K) if s % eng.cfg["entire_A_every"] == 0)
    roof = None
    if args.prof_kernel and prof_n.value:
        # algorithmic FLOPs of ONE launch of the timed kernel: every ViT forward launch covers 2 passes x T tokens
        # (targets A', B' on one stream, generated x', y' on the other; the entire-image branch is a 2-pass batch too)
        P = 2
        shapes = {1: ("gemm_nt_kernel<128,128,BIAS|GELU|OUT_BF,2> (fc1 fwd)", 2.0 * P * T * hidden * D),
                  2: ("gemm_nt_kernel<128,64,BIAS|OUT_BF|OUT_T,2> (qkv fwd)", 2.0 * P * T * 3 * D * D),
                  3: ("attn_fwd_kernel<1>", 4.0 * P * T * T * D),
                  4: ("gemm_nt_kernel<64,64,BIAS|RESID|OUT_F32,4> (fc2 fwd)", 2.0 * P * T * hidden * D)}
        kname, flops = shapes[args.prof_kernel]
        raw_ms = prof_ms.value / prof_n.value
        avg_ms = max(raw_ms - ev_overhead_ms, 1e-6)
        ach = flops / (avg_ms * 1e-3) / 1e12
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "roofline_traffic.json")
        if os.path.exists(tfile) and args.size == 224 and args.model == "dino_vitb8":
            try:
                traffic = json.load(open(tfile)).get(str(args.prof_kernel))
            except Exception:
                traffic = None
        roof = {"bound": "mfma", "kernel": kname, "achieved": round(ach, 2), "peak": 2500.0, "unit": "TFLOP/s",
                "frac": round(ach / 2500.0, 4), "traffic": traffic,
                "traffic_source": None if traffic is None else "static: profiles/roofline_traffic.json (PMC passes of an earlier run, not re-measured by this command)",
                "avg_launch_us": round(avg_ms * 1e3, 2), "event_pair_overhead_us": round(ev_overhead_ms * 1e3, 2),
                "launches": prof_n.value,
                "note": "algorithmic FLOPs of one launch (2 passes x T tokens) / mean HIP-event duration of its launches, "
                        "measured on the launch stream over the instrumented continuation of the timed steps (the timed "
                        "region itself replays hipGraphs), minus the median cost of an empty event pair"}
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
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"Splice pair {hw[0]}x{hw[1]}, {args.model} (T={T}), 1 pair per GPU, "
                               f"{n_entire} of {K} timed steps include the entire-image branch",
                   "pairs": world, "pairs_per_hour_at_2000_steps": round(value * 3600 / 2000, 2),
                   "per_rank_steps_per_s": [round(K / t, 2) for t in per_rank_elapsed],
                   "generator_dtype": "f32", "last_loss": round(losses["loss"], 5)},
        "roofline": roof, "cpu_baseline": cpu,
    }
    print(json.dumps(out), flush=True)
    rep.close()


if __name__ == "__main__":
    main()
