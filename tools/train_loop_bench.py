#!/usr/bin/env python3
"""End-to-end train_model() rate on a synthetic 224x224 pair with the reference's default config (random >= 95 % crops,
use_augmentations as configured, PNG + callback every log_images_freq steps).  Run on the GPU box."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
os.environ.setdefault("SPLICE_SYNTHETIC_WEIGHTS", "1")
from PIL import Image
from splice_amd import synth
from splice_amd.train import train_model

d = tempfile.mkdtemp()
A, B = synth.smooth_image_pair(5, 0, 224, 224)
for name, img in (("A", A), ("B", B)):
    os.makedirs(os.path.join(d, name))
    Image.fromarray((img.transpose(1, 2, 0) * 255).astype(np.uint8)).save(os.path.join(d, name, "img.png"))
for aug in (False, True):
    n = 400
    FREQ = int(os.environ.get("LOG_FREQ", "100"))
    t = {}
    def cb(img, t=t):
        t.setdefault("first", time.perf_counter())
        t["last"] = time.perf_counter(); t["n"] = t.get("n", 0) + 1
    train_model(d, callback=cb, cfg_overrides=dict(n_epochs=n, log_images_freq=FREQ, use_augmentations=aug, seed=1), progress=False)
    # callbacks fire every FREQ steps: n - FREQ steps between the first and the last
    rate = (n - FREQ) / (t["last"] - t["first"])
    print(f"use_augmentations={aug}: {rate:.1f} steps/s end to end (random crops, PNG every {FREQ} steps)")
