#!/usr/bin/env python3
"""End-to-end wall clock of train_model at the reference's DEFAULT workload shape: a synthetic 1200 x 900 PNG pair on disk (conf/default/config.yaml: A_resize -1, global
crops of 95 .. 100 % of the short side resized to 224, entire image every 75 steps, augmentations, a logged 1200 x 900 image every 10 steps), E2E_STEPS steps (default 1000;
the reference's default is 10000), first and second pair of a process.  Prints seconds per 1000 steps including image decode, engine construction and the PNG writer."""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PIL import Image
from splice_amd import synth
from splice_amd.train import train_model

STEPS = int(os.environ.get("E2E_STEPS", "1000"))
os.environ["SPLICE_SYNTHETIC_WEIGHTS"] = "1"
root = tempfile.mkdtemp()
dirs = []
for i in range(2):
    A, B = synth.smooth_image_pair(91, i, 900, 1200)
    for side, img in (("A", A), ("B", B)):
        d = os.path.join(root, f"p{i}", side)
        os.makedirs(d)
        Image.fromarray((img.transpose(1, 2, 0) * 255).astype(np.uint8)).save(os.path.join(d, "img.png"))
    dirs.append(os.path.join(root, f"p{i}"))
for i, d in enumerate(dirs):
    t0 = time.perf_counter()
    train_model(d, cfg_overrides=dict(n_epochs=STEPS, seed=1), progress=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"train_model at 900 x 1200, pair {i} of the process, {STEPS} steps: {dt:.2f} s = {dt / STEPS * 1e3:.2f} ms per step end to end "
          f"({STEPS / dt:.1f} steps/s; a 10000-step default run: {dt / STEPS * 1e4 / 60:.1f} min)", flush=True)
    assert os.path.exists(os.path.join(d, "out", "output.png"))
