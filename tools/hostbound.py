import sys, time
sys.path.insert(0, "/root/repo")
import torch
from splice_amd.engine import synthetic_engine
cfg = dict(dino_model_name="dino_vitb8", dino_global_patch_size=224)
eng, A, B = synthetic_engine(cfg, pair_id=0, hw=(224, 224), seed=1234)
for _ in range(30): eng.step(A, B, A)
torch.cuda.synchronize()
for K in (50, 200):
    t0 = time.perf_counter()
    for _ in range(K): eng.step(A, B, A)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"K={K}: enqueue {1e3*(t1-t0)/K:.3f} ms/step, total {1e3*(t2-t0)/K:.3f} ms/step")
for K in (1, 2, 5, 10, 20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K): eng.step(A, B, A)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"after sync, K={K}: enqueue {1e3*(t1-t0)/K:.3f} ms/step, total {1e3*(t2-t0)/K:.3f} ms/step")
