import sys, time, numpy as np
sys.path.insert(0, ".")
import torch, plspy_amd
X = np.random.RandomState(0).randn(60, 200_000)
Y = np.random.RandomState(1).randn(60, 4)
for method, kw in (("mct", {}), ("rb", {"Y": Y})):
    for rep in range(2):
        np.random.seed(3); torch.cuda.synchronize(); t0 = time.perf_counter()
        r = plspy_amd.PLS(X, [10, 10], 3, num_perm=0, num_boot=0, num_split=500, lv=2, pls_method=method, **kw)
        torch.cuda.synchronize(); print(method, "500 splits (tt + sh):", round(time.perf_counter() - t0, 3), "s")
