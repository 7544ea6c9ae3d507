"""Where a warm plspy_amd.PLS() call at config 2 spends its 10 ms: wall time with X on the host and with
X already on the device (as plspy_amd.io leaves it), then a cProfile of one warm call of each."""
import cProfile
import pstats
import sys
import time

import numpy as np
import torch

import plspy_amd

X = np.random.RandomState(0).randn(60, 200_000)
Xd = torch.as_tensor(X).cuda()
for name, x in (("host", X), ("device", Xd)):
    ts = []
    for _ in range(8):
        np.random.seed(1234)
        t0 = time.perf_counter()
        plspy_amd.PLS(x, [10, 10], 3, num_perm=1000, num_boot=1000, pls_method="mct")
        ts.append(time.perf_counter() - t0)
    print(name, " ".join(f"{t * 1e3:.2f}" for t in ts), "ms ->", f"{2000 / min(ts[1:]):.0f} resamples/s")
    pr = cProfile.Profile()
    np.random.seed(1234)
    pr.enable()
    plspy_amd.PLS(x, [10, 10], 3, num_perm=1000, num_boot=1000, pls_method="mct")
    pr.disable()
    pstats.Stats(pr, stream=sys.stdout).sort_stats("tottime").print_stats(18)
