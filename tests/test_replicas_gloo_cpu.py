"""CPU, world_size 2 over gloo: the N>1 path of bench.py (process-per-GPU replicas of independent
pairs) -- rank->pair mapping, barrier, max-over-ranks timing, aggregate throughput; and that two
replicas' synthetic workloads are different pairs with identical (replicated) ViT weights."""
import os
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from splice_amd import synth
    from splice_amd.dist import Replicas, aggregate_throughput
    rep = Replicas(backend="gloo")
    pid = rep.pair_id()
    A, B = synth.image_pair(1234, pid, 8, 8)
    vit_sum = float(synth.vit_params(1234, patch=8, dim=64, depth=1, img_size=16)["blocks.0.attn.qkv.weight"].sum())
    rep.barrier()
    elapsed = 1.0 + rank          # pretend rank 1 is slower
    tmax = rep.max_over_ranks(elapsed)
    sums = rep.gather_floats(float(A.sum()))
    vits = rep.gather_floats(vit_sum)
    out_q.put((rank, pid, tmax, aggregate_throughput(100, world, tmax), sums, vits))
    rep.close()


def test_two_replicas_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, p0, t0, v0, s0, w0), (r1, p1, t1, v1, s1, w1) = res
    assert (r0, r1) == (0, 1) and (p0, p1) == (0, 1)          # one distinct pair per rank
    assert t0 == t1 == 2.0                                    # max over ranks
    assert v0 == v1 == 100 * 2 / 2.0                          # whole-job steps / max time
    assert s0 == s1 and abs(s0[0] - s0[1]) > 1e-3             # different pairs
    assert w0 == w1 and w0[0] == w0[1]                        # same (replicated) frozen ViT weights


def test_single_process_replicas():
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)
    from splice_amd.dist import Replicas
    rep = Replicas()
    assert rep.world == 1 and rep.pair_id() == 0
    rep.barrier()
    assert rep.max_over_ranks(3.5) == 3.5
