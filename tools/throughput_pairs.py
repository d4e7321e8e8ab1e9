#!/usr/bin/env python3
"""Throughput mode study: P independent pairs optimised concurrently on ONE GPU (each engine on its own stream pair,
frozen ViT weights shared).  A single pair's step is a latency-bound chain of ~500 launches; independent pairs fill the
idle CUs.  Run on the GPU box:  python tools/throughput_pairs.py [P ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from splice_amd.engine import synthetic_engine

Ps = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4]
cfg = dict(dino_model_name="dino_vitb8", dino_global_patch_size=224)
K, W = 150, 15
vit = None
for P in Ps:
    engs = []
    for i in range(P):
        eng, A, B = synthetic_engine(cfg, pair_id=i, hw=(224, 224), seed=1234, vit_engine=vit)
        vit = eng.vit
        engs.append((eng, A, B, torch.cuda.Stream()))
    torch.cuda.synchronize()

    def run(n):
        for _ in range(n):
            for eng, A, B, st in engs:
                with torch.cuda.stream(st):
                    eng.step(A, B, A)
    run(W)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(K)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"P={P}: {P * K / el:7.1f} steps/s aggregate ({K / el:6.1f} per pair, {el / K * 1e3:.2f} ms per round), "
          f"{P * K / el * 3600 / 2000:.0f} pairs/h at 2000 steps", flush=True)
    del engs
