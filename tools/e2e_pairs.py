#!/usr/bin/env python3
"""End-to-end wall clock of BASELINE configs[1] / [2]-per-GPU through the product entry points: K synthetic 224x224 pairs written
as PNGs, optimised for 2000 steps each (reference defaults otherwise: random >= 95 % crops, augmentations, a logged image every
10 steps, PNG written by the worker thread) -- one pair at a time (train_model) and P pairs in the same launches
(train_pairs).  Prints pairs/hr including image decode, engine construction, graph capture and PNG encode."""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PIL import Image
from splice_amd import synth
from splice_amd.train import train_model, train_pairs

K = int(os.environ.get("E2E_PAIRS", "8"))
STEPS = int(os.environ.get("E2E_STEPS", "2000"))
os.environ["SPLICE_SYNTHETIC_WEIGHTS"] = "1"
root = tempfile.mkdtemp()
dirs = []
for i in range(K):
    A, B = synth.smooth_image_pair(77, i, 224, 224)
    for side, img in (("A", A), ("B", B)):
        d = os.path.join(root, f"p{i}", side)
        os.makedirs(d)
        Image.fromarray((img.transpose(1, 2, 0) * 255).astype(np.uint8)).save(os.path.join(d, "img.png"))
    dirs.append(os.path.join(root, f"p{i}"))
over = dict(n_epochs=STEPS, seed=1)
t0 = time.perf_counter()
train_model(dirs[0], cfg_overrides=over, progress=False)
torch.cuda.synchronize()
t1 = time.perf_counter() - t0
print(f"train_model, first pair of the process x {STEPS} steps (incl. ViT weights: 2 s): {t1:.2f} s -> {3600 / t1:.0f} pairs/hr ({STEPS / t1:.1f} steps/s end to end)", flush=True)
t0 = time.perf_counter()
train_model(dirs[1], cfg_overrides=over, progress=False)
torch.cuda.synchronize()
t1 = time.perf_counter() - t0
print(f"train_model, next pair of the same process (a batch worker's steady state): {t1:.2f} s -> {3600 / t1:.0f} pairs/hr ({STEPS / t1:.1f} steps/s end to end)", flush=True)
for P in (4, 8):
    if P > K:
        continue
    t0 = time.perf_counter()
    train_pairs(dirs[:P], cfg_overrides=over, progress=False)
    torch.cuda.synchronize()
    tp = time.perf_counter() - t0
    print(f"train_pairs, {P} pairs x {STEPS} steps in the same launches: {tp:.2f} s -> {3600 * P / tp:.0f} pairs/hr ({P * STEPS / tp:.1f} pair-steps/s end to end)", flush=True)
