#!/usr/bin/env python3
"""cProfile of a short train_model run (host side): which Python-level calls the default-config loop spends its time in."""
import cProfile, os, pstats, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image
os.environ["SPLICE_SYNTHETIC_WEIGHTS"] = "1"
from splice_amd import synth
from splice_amd.train import train_model
root = tempfile.mkdtemp()
A, B = synth.smooth_image_pair(77, 0, 224, 224)
for side, img in (("A", A), ("B", B)):
    os.makedirs(os.path.join(root, side))
    Image.fromarray((img.transpose(1, 2, 0) * 255).astype(np.uint8)).save(os.path.join(root, side, "img.png"))
vs = synth.vit_params(1234, "dino_vitb8", img_size=224)
train_model(root, cfg_overrides=dict(n_epochs=50, seed=1), vit_state=vs, progress=False)   # warm-up
pr = cProfile.Profile()
pr.enable()
train_model(root, cfg_overrides=dict(n_epochs=int(os.environ.get("E2E_STEPS", "800")), seed=1), vit_state=vs, progress=False)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(32)
