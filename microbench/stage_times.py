"""Per-call wall time of the stages of warm PLS() calls at config 2 (which stage makes some calls slow?)."""
import sys, time, functools, collections
import numpy as np
sys.path.insert(0, ".")
import torch
import plspy_amd
from plspy_amd import engine, bootstrap_permutation as bp, resample
acc = collections.OrderedDict()
def wrap(obj, name, label=None):
    f = getattr(obj, name)
    label = label or name
    @functools.wraps(f)
    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[label] = acc.get(label, 0.0) + (time.perf_counter() - t0) * 1e3
    setattr(obj, name, g)
E = engine.ProjectionEngine
for n in ("__init__", "thin_svd", "apply_operator", "boot_phase", "perm_phase", "boot_finalize", "dev"):
    wrap(E, n, "eng." + n)
wrap(torch.Tensor, "cpu", "Tensor.cpu")
wrap(torch.Tensor, "to", "Tensor.to")
R = bp._ResampleTestPLS
for n in ("__init__", "_bootstrap_test_start", "_bootstrap_test_finish"):
    if hasattr(R, n):
        wrap(R, n, "rt." + n)
wrap(resample, "task_permutations"); wrap(resample, "bootstraps")
X = np.random.RandomState(0).randn(60, 200_000)
def run():
    np.random.seed(1)
    return plspy_amd.PLS(X, (10, 10), 3, num_perm=1000, num_boot=1000, pls_method="mct")
run(); run()
for i in range(8):
    acc.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = run(); torch.cuda.synchronize()
    w = (time.perf_counter() - t0) * 1e3
    print(f"wall {w:6.1f} | " + " ".join(f"{k}={v:.1f}" for k, v in acc.items()), flush=True)
