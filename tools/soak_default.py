#!/usr/bin/env python3
"""Soak of what the reference actually runs: `train_model` with the reference's default configuration -- 1200 x 900 images,
random 95 .. 100 % crops resized to 224, the entire image every 75 steps, augmentations, a logged image every 10 steps
(conf/default/config.yaml:3,5-7) -- for SOAK_STEPS steps per run (default 10000 = config.yaml:29), SOAK_RUNS runs back to back in ONE
long-lived process (the situation of a batch worker, where round 4's crash inside the runtime's completion handler happened).

Per run: wall seconds, steps/s, the final loss, device memory in use, and the step handle's graph statistics (captures that updated a pooled
executable in place / updates the runtime refused [executable parked for good] / executables instantiated).  A fault would end the
process: the last line printed names the run.  SOAK_SIZE=HxW (default 900x1200), SOAK_PAIRS distinct image pairs cycled through."""
import ctypes as C
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PIL import Image

from splice_amd import _lib, synth
from splice_amd.train import train_model

STEPS = int(os.environ.get("SOAK_STEPS", "10000"))
RUNS = int(os.environ.get("SOAK_RUNS", "4"))
H, W = (int(x) for x in os.environ.get("SOAK_SIZE", "900x1200").lower().split("x"))
NP = int(os.environ.get("SOAK_PAIRS", "2"))
os.environ["SPLICE_SYNTHETIC_WEIGHTS"] = "1"
root = tempfile.mkdtemp()
dirs = []
for i in range(NP):
    A, B = synth.smooth_image_pair(91, i, H, W)
    for side, img in (("A", A), ("B", B)):
        d = os.path.join(root, f"p{i}", side)
        os.makedirs(d)
        Image.fromarray((img.transpose(1, 2, 0) * 255).astype(np.uint8)).save(os.path.join(d, "img.png"))
    dirs.append(os.path.join(root, f"p{i}"))
print(f"soak: {RUNS} runs x {STEPS} steps of train_model at {H} x {W} (reference default config), one process, device {torch.cuda.get_device_name(0)}", flush=True)
tot_ref = tot_inst = 0
t_all = time.perf_counter()
for r in range(RUNS):
    d = dirs[r % NP]
    t0 = time.perf_counter()
    eng = train_model(d, cfg_overrides=dict(n_epochs=STEPS, seed=1 + r), progress=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = (C.c_longlong * 3)()
    _lib.check(_lib.lib().splice_step_graph_stats(eng.handle, st))
    free, total = torch.cuda.mem_get_info()
    loss = eng.losses()["loss"]
    tot_ref += st[1]
    tot_inst += st[2]
    print(f"run {r}: {dt:7.1f} s, {STEPS / dt:6.1f} steps/s end to end, final loss {loss:.4f}, device memory in use {(total - free) / 2**20:.0f} MiB, "
          f"graph captures: updated in place {st[0]}, refused {st[1]}, instantiated {st[2]}", flush=True)
    assert np.isfinite(loss) and os.path.exists(os.path.join(d, "out", "output.png"))
    del eng
print(f"soak done: {RUNS} runs x {STEPS} steps without a fault in {time.perf_counter() - t_all:.0f} s; update refusals {tot_ref}, executables instantiated {tot_inst} "
      f"(none destroyed: the library has no hipGraphExecDestroy call)", flush=True)
