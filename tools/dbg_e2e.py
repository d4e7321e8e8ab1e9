#!/usr/bin/env python3
"""Teardown / re-entry stress of the product entry point: N train_model runs in ONE process (a batch worker's life), optionally
under cProfile, with faulthandler on.  E2E_STEPS (300), E2E_RUNS (6), E2E_CPROFILE (0 / 1)."""
import os, sys, time, tempfile, cProfile, gc, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
from splice_amd import synth
from splice_amd.train import train_model
os.environ["SPLICE_SYNTHETIC_WEIGHTS"] = "1"
_bt = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", "segv_bt.so")
if os.path.exists(_bt):   # native backtrace of a crashing thread (tools/micro/segv_bt.c)
    ctypes.CDLL(_bt).segv_bt_install()
else:
    import faulthandler
    faulthandler.enable()
R = int(os.environ.get("E2E_RUNS", "6"))
N = int(os.environ.get("E2E_STEPS", "300"))
root = tempfile.mkdtemp(); dirs = []
for i in range(R):
    A, B = synth.smooth_image_pair(77, i, 224, 224)
    for side, img in (("A", A), ("B", B)):
        d = os.path.join(root, f"p{i}", side); os.makedirs(d)
        Image.fromarray((img.transpose(1, 2, 0) * 255).astype(np.uint8)).save(os.path.join(d, "img.png"))
    dirs.append(os.path.join(root, f"p{i}"))
over = dict(n_epochs=N, seed=1)
pr = cProfile.Profile() if os.environ.get("E2E_CPROFILE") == "1" else None
for k in range(R):
    if pr is not None and k == R - 1:
        pr.enable()
    t0 = time.perf_counter(); train_model(dirs[k], cfg_overrides=over, progress=False); torch.cuda.synchronize(); print("pair", k, round(time.perf_counter() - t0, 3), flush=True)
if pr is not None:
    pr.disable()
gc.collect()
print("done", flush=True)
