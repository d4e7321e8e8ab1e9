#!/usr/bin/env python3
"""Two MultiPairEngines (P pairs each, one shared frozen ViT) stepped from two host streams of ONE process against one engine of 2P pairs:
does staggering one group's generator phase against the other's ViT phase beat the bigger batch?  python tools/two_engines.py [P] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from splice_amd.engine import synthetic_engine
P = int(sys.argv[1]) if len(sys.argv) > 1 else 4
K = int(sys.argv[2]) if len(sys.argv) > 2 else 100
cfg = dict(dino_model_name="dino_vitb8", dino_global_patch_size=224)
big, A2, B2 = synthetic_engine(cfg, pair_id=0, pairs=2 * P)
e1, A, B = synthetic_engine(cfg, pair_id=0, pairs=P, vit_engine=big.vit)
e2, Ab, Bb = synthetic_engine(cfg, pair_id=P, pairs=P, vit_engine=big.vit)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run_big(n):
    for _ in range(n): big.step(A2, B2, A2)
def run_two(n, lag):
    for i in range(n):
        with torch.cuda.stream(s1): e1.step(A, B, A)
        with torch.cuda.stream(s2): e2.step(Ab, Bb, Ab)
for name, fn in (("one engine of %d pairs" % (2 * P), lambda n: run_big(n)), ("two engines of %d pairs on two streams" % P, lambda n: run_two(n, 0))):
    fn(15); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(K); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{name}: {dt / K * 1e3:.3f} ms per round of {2 * P} pair-steps -> {2 * P * K / dt:.1f} pair-steps/s", flush=True)
