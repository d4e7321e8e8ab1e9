import sys, time
sys.path.insert(0, "/root/repo")
import torch, numpy as np
from splice_amd.engine import synthetic_engine
cfg = dict(dino_model_name="dino_vitb8", dino_global_patch_size=224)
eng, A, B = synthetic_engine(cfg, pair_id=0, hw=(224, 224), seed=1234)
rng = np.random.RandomState(0)
def mem(): 
    torch.cuda.synchronize(); f, t = torch.cuda.mem_get_info(); return (t - f) / 2**20
for phase in range(4):
    t0 = time.perf_counter()
    for i in range(1500):
        if phase % 2 == 0:
            eng.step(A, B, A)
        else:
            sa, sb = rng.randint(213, 225), rng.randint(213, 225)
            eng.step(A[:, :sa, :sa].contiguous(), B[:, :sb, :sb].contiguous(), A)
    m = mem()
    print(f"phase {phase} ({'fixed' if phase % 2 == 0 else 'random'} crops): {1500 / (time.perf_counter() - t0):.1f} steps/s, device memory in use {m:.0f} MiB, loss {eng.losses()['loss']:.4f}", flush=True)

import ctypes as C
from splice_amd import _lib
st = (C.c_longlong * 3)()
_lib.check(_lib.lib().splice_step_graph_stats(eng.handle, st))
print("graph executables of this handle: updated in place %d, updates refused %d, instantiated %d" % tuple(st), flush=True)
