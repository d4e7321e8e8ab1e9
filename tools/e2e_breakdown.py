#!/usr/bin/env python3
"""Where the end-to-end time of one train_model run goes (setup vs loop): run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
t0 = time.perf_counter()
from splice_amd import synth, _lib
from splice_amd.engine import SpliceEngine
from splice_amd.vit import VitEngine
from splice_amd.networks import define_G
import yaml
_lib.lib()
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
def lap(msg):
    global t0
    torch.cuda.synchronize(); t = time.perf_counter(); print(f"{msg:55s} {t - t0:7.3f} s", flush=True); t0 = t
lap("imports + library load + HIP context")
vs = synth.vit_params(1234, "dino_vitb8", img_size=224); lap("synthetic ViT-B/8 weights on the CPU (stands in for torch.load)")
cfg = yaml.safe_load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "splice_amd", "conf", "default", "config.yaml")))
vit = VitEngine("dino_vitb8", device=torch.device("cuda")).load_state_dict(vs); lap("VitEngine: upload + bf16 packing")
netG = define_G(cfg['init_type'], cfg['init_gain'], device=torch.device("cuda")); lap("define_G")
gs = {k: v.detach() for k, v in netG.state_dict().items() if k in netG.engine.table}
eng = SpliceEngine(cfg, vs, gs, (224, 224), (224, 224), device=torch.device("cuda"), vit_engine=vit); lap("SpliceEngine (plans, workspaces) with a shared VitEngine")
A = torch.rand(3, 224, 224, device="cuda"); B = torch.rand(3, 224, 224, device="cuda")
eng.step(A[None], B[None], A[None]); lap("first step (eager + capture)")
for _ in range(10): eng.step(A[None], B[None], A[None])
lap("10 more steps (graph capture of the ordinary regime)")
for _ in range(500): eng.step(A[None], B[None], A[None])
lap("500 steps")
