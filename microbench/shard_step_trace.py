"""A rank's step of config 2 on N GPUs (R/N permutations + R/N bootstraps, no collectives), a few dozen times:
run under `rocprofv3 --kernel-trace` to see what a step consists of besides the two projection kernels.
Usage: python microbench/shard_step_trace.py [N] [steps]"""
import sys
import numpy as np
sys.path.insert(0, ".")
import torch
from plspy_amd import dist, operators, resample
from plspy_amd.engine import ProjectionEngine

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
co = np.array([[10] * 3, [10] * 3])
X = np.random.RandomState(0).randn(60, 200_000)
W = operators.mean_centre_operator(co, 0)
Wm = operators.cell_mean_operator(co)
U, s, Vt = np.linalg.svd(W @ X, full_matrices=False)
V = Vt.T
k = U.shape[1]
eng = ProjectionEngine(X)
M = W.T @ U
ref = eng.dev(V * s)
Xm = eng.apply_operator(Wm)
Md = eng.dev(M)
np.random.seed(1234)
plo, phi = dist.shard_bounds(1000, 0, N)
d_perm = eng.dev(resample.task_permutations(co, 1000)[plo:phi], torch.int32)
d_boot = eng.dev(resample.bootstraps(co, 1000)[plo:phi], torch.int32)


def step():
    prep = eng.perm_prepare(k, d_perm, Md)
    res = eng.boot_phase(k, inds=d_boot, M=Md, ref=ref, Xm=Xm, overlap_tail=True)
    ssq = eng.perm_phase(k, inds=d_perm, M=Md, prepared=prep)
    with eng.tail_stream():
        out = eng.boot_finalize(res["S12"][0], res["S12"][1], phi - plo, num=ref)
    eng.join()
    return out, ssq


for _ in range(10):
    step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(steps):
    step()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"N={N}: {1e3 * (time.perf_counter() - t0) / steps:.4f} ms per step; the host needs {1e3 * t_host / steps:.4f} ms to enqueue one")
