import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import plspy_amd
from plspy_amd import engine as E
from plspy_amd.bootstrap_permutation import ResampleTest
X = np.random.RandomState(0).randn(120, 200_000)
Y = np.random.RandomState(1).randn(120, 8)
co = np.array([[20] * 3, [20] * 3])
np.random.seed(1234)
res = plspy_amd.PLS(X, [20, 20], 3, Y=Y, num_perm=0, num_boot=0, pls_method="rb")
U, s, V = res.V, res.s.copy(), res.U
eng = E.ProjectionEngine(X)
for thr in (32 << 20, 1 << 60, 32 << 20, 1 << 60):   # pageable (pre-faulted) results / page-locked views
    E.PINNED_COPY_BYTES = thr
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ResampleTest._create("rb", X, Y, U, s.copy(), V, co, None, nperm=0, nboot=2000, lvcorrs_orig=res.lvcorrs, engine=eng)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print("copy threshold", thr >> 20, "MiB:", [round(t * 1e3, 1) for t in ts])
