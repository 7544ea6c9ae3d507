"""Duration of the two register-resident projection kernels of config 2 against the number of resamples of a launch
(hipEvent durations through plsr_timing_*)."""
import ctypes, sys
import numpy as np
sys.path.insert(0, ".")
import torch
from plspy_amd import operators, resample
from plspy_amd.engine import ProjectionEngine
co = np.array([[10] * 3, [10] * 3])
X = np.random.RandomState(0).randn(60, 200_000)
W = operators.mean_centre_operator(co, 0)
Wm = operators.cell_mean_operator(co)
U, s, Vt = np.linalg.svd(W @ X, full_matrices=False)
k = U.shape[1]
eng = ProjectionEngine(X)
ref = eng.dev(Vt.T * s); Xm = eng.apply_operator(Wm); Md = eng.dev(W.T @ U)
np.random.seed(1)
P, B = resample.task_permutations(co, 1000), resample.bootstraps(co, 1000)
lib = eng.lib
for R in (16, 32, 64, 125, 250, 500, 1000):
    dp, db = eng.dev(P[:R], torch.int32), eng.dev(B[:R], torch.int32)
    for _ in range(3):
        eng.boot_phase(k, inds=db, M=Md, ref=ref, Xm=Xm); eng.perm_phase(k, inds=dp, M=Md)
    torch.cuda.synchronize()
    lib.plsr_timing_enable(1)
    for _ in range(20):
        eng.boot_phase(k, inds=db, M=Md, ref=ref, Xm=Xm); eng.perm_phase(k, inds=dp, M=Md)
    torch.cuda.synchronize()
    lib.plsr_timing_enable(0)
    ms = (ctypes.c_double * 200)(); kind = (ctypes.c_int32 * 200)()
    nt = lib.plsr_timing_collect(ms, kind, 200)
    b = np.median([ms[i] for i in range(nt) if kind[i] == 1]); p_ = np.median([ms[i] for i in range(nt) if kind[i] == 0])
    print(f"R={R:5d}: boot {b:.4f} ms ({1e3 * b / R:.2f} us/resample)  perm {p_:.4f} ms ({1e3 * p_ / R:.2f} us/resample)", flush=True)
