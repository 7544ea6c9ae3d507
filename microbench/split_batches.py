"""Config 4's split-half phases with the shard's items sent to the device in 1 / 2 / 4 batches."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
from plspy_amd import split_half_resampling as sh
from plspy_amd.engine import ProjectionEngine
X = np.random.RandomState(0).randn(120, 200_000)
Y = np.random.RandomState(1).randn(120, 8)
co = np.array([[20] * 3, [20] * 3])
eng = ProjectionEngine(X)
kw = dict(mctype=0, bscan=[1, 2], engine=eng)
np.random.seed(1)
for rep in range(2):
    for nb in (1, 2, 4, 8):
        sh.DECOMPOSE_BATCHES = nb
        ts = []
        for fn in (lambda: sh.split_half_test_train("mb", X, Y, co, 1000, **kw), lambda: sh.split_half("mb", X, Y, co, 1000, lv=2, CI=0.95, **kw)):
            fn()
            torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        print(f"batches {nb}: test_train {ts[0] * 1e3:6.1f} ms, split_half {ts[1] * 1e3:6.1f} ms -> {1000 / sum(ts):6.0f} splits/s", flush=True)
