import sys, time, os
sys.path.insert(0, "/root/repo")
import torch, numpy as np
from splice_amd.engine import synthetic_engine
from splice_amd import _lib
cfg = dict(dino_model_name="dino_vitb8", dino_global_patch_size=224)
eng, A, B = synthetic_engine(cfg, pair_id=0, hw=(224, 224), seed=1234)
def run(n, sizes, graph=1):
    _lib.check(_lib.lib().splice_step_use_graph(eng.handle, graph))
    rng = np.random.RandomState(0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        sa, sb = rng.choice(sizes), rng.choice(sizes)
        eng.step(A[:, :sa, :sa].contiguous(), B[:, :sb, :sb].contiguous(), A)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
run(20, [224])
print("fixed crops, graph : %.2f ms/step" % run(100, [224]))
print("fixed crops, eager : %.2f ms/step" % run(100, [224], graph=0))
print("random crops 213..224, graph (recapture on change): %.2f ms/step" % run(60, list(range(213, 225))))
print("random crops 213..224, eager: %.2f ms/step" % run(60, list(range(213, 225)), graph=0))
