"""Three warm plspy_amd.PLS(...) calls at config 2 (run under rocprofv3 --kernel-trace)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
import plspy_amd
X = np.random.RandomState(0).randn(60, 200_000)
def run():
    np.random.seed(1)
    return plspy_amd.PLS(X, (10, 10), 3, num_perm=1000, num_boot=1000, pls_method="mct")
run()
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); run(); torch.cuda.synchronize()
    print("wall ms", (time.perf_counter() - t0) * 1e3, flush=True)
