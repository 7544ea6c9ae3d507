"""K5i alone at config 3's shape: 125 bootstrap samples of X (120 x 200 000), k = 48; time per item.
PLSR_LIB selects a developer (ablation) build."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
from plspy_amd.engine import ProjectionEngine
n, p, k, items = 120, 200_000, 48, 125
rs = np.random.RandomState(0)
eng = ProjectionEngine(rs.randn(n, p))
vs = torch.randn((items, k, p), dtype=torch.float64, device=eng.device)
bounds = np.arange(0, n + 1, 20)
idx = np.concatenate([rs.randint(l, h, size=(items, h - l)) for l, h in zip(bounds[:-1], bounds[1:])], axis=1).astype(np.int32)
d_idx = eng.dev(idx, torch.int32)
L = torch.empty((items, k, n), dtype=torch.float64, device=eng.device)
Zt = torch.empty((items, k, n), dtype=torch.float64, device=eng.device)
modes = (("index", lambda: eng.latent_batch_index(vs, n, idx, d_idx, L, None)),
         ("full ", lambda: eng.latent_batch(vs, n, Zt, None)))
for name, fn in modes[:1] if len(sys.argv) > 1 else modes:
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record(); torch.cuda.synchronize()
    print(name, "us per item: %.2f" % (e0.elapsed_time(e1) * 1e3 / 5 / items), "distinct rows max", eng.distinct_rows(idx), flush=True)
