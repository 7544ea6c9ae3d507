"""PLS() end to end at config 5 (240 x 500 000, 5000 + 5000) with large results as pre-faulted pageable arrays
(default) and as views of page-locked buffers."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import plspy_amd
from plspy_amd import engine as E
X = np.random.RandomState(0).randn(240, 500_000)
for thr in (32 << 20, 1 << 60, 32 << 20, 1 << 60):
    E.PINNED_COPY_BYTES = thr
    ts = []
    for _ in range(3):
        np.random.seed(1234)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        plspy_amd.PLS(X, [20] * 4, 3, num_perm=5000, num_boot=5000, pls_method="mct")
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print("threshold", thr >> 20, "MiB:", [round(t, 3) for t in ts], flush=True)
