import os, sys, time, tempfile, cProfile, pstats
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from PIL import Image
from splice_amd import synth
from splice_amd.train import train_model
os.environ["SPLICE_SYNTHETIC_WEIGHTS"] = "1"
root = tempfile.mkdtemp(); dirs = []
for i in range(3):
    A, B = synth.smooth_image_pair(77, i, 224, 224)
    for side, img in (("A", A), ("B", B)):
        d = os.path.join(root, f"p{i}", side); os.makedirs(d)
        Image.fromarray((img.transpose(1, 2, 0) * 255).astype(np.uint8)).save(os.path.join(d, "img.png"))
    dirs.append(os.path.join(root, f"p{i}"))
over = dict(n_epochs=2000, seed=1)
train_model(dirs[0], cfg_overrides=over, progress=False); torch.cuda.synchronize()
t0 = time.perf_counter(); train_model(dirs[1], cfg_overrides=over, progress=False); torch.cuda.synchronize(); print("steady pair", time.perf_counter() - t0)
pr = cProfile.Profile(); pr.enable()
train_model(dirs[2], cfg_overrides=over, progress=False); torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
