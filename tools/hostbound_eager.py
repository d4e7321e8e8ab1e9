"""Is the EAGER step (train_model's regime under random crop sizes) bound by the host or by the GPU?  Host time to enqueue K steps
against the time until the GPU has finished them, eager and graph form (run on the GPU box)."""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
from splice_amd.engine import synthetic_engine
from splice_amd import _lib
cfg = dict(dino_model_name="dino_vitb8", dino_global_patch_size=224)
eng, A, B = synthetic_engine(cfg, pair_id=0, hw=(224, 224), seed=1234)
for graph in (1, 0, 1, 0):
    _lib.check(_lib.lib().splice_step_use_graph(eng.handle, graph))
    for _ in range(30): eng.step(A, B, A)
    torch.cuda.synchronize()
    for K in (5, 200):
        t0 = time.perf_counter()
        for _ in range(K): eng.step(A, B, A)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"graph={graph} K={K}: enqueue {1e3*(t1-t0)/K:.3f} ms/step, until done {1e3*(t2-t0)/K:.3f} ms/step", flush=True)
