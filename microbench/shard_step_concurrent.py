"""A rank's step of config 2 on N GPUs with the permutation kernel on a second stream beside the bootstrap kernel
(they are independent), against the serial order.  Usage: python microbench/shard_step_concurrent.py [N] [steps]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
from plspy_amd import dist, operators, resample
from plspy_amd.engine import ProjectionEngine

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
co = np.array([[10] * 3, [10] * 3])
X = np.random.RandomState(0).randn(60, 200_000)
W = operators.mean_centre_operator(co, 0)
Wm = operators.cell_mean_operator(co)
U, s, Vt = np.linalg.svd(W @ X, full_matrices=False)
k = U.shape[1]
eng = ProjectionEngine(X)
ref = eng.dev(Vt.T * s)
Xm = eng.apply_operator(Wm)
Md = eng.dev(W.T @ U)
np.random.seed(1234)
plo, phi = dist.shard_bounds(1000, 0, N)
d_perm = eng.dev(resample.task_permutations(co, 1000)[plo:phi], torch.int32)
d_boot = eng.dev(resample.bootstraps(co, 1000)[plo:phi], torch.int32)
side = torch.cuda.Stream()


def step(concurrent):
    main = torch.cuda.current_stream()
    prep = eng.perm_prepare(k, d_perm, Md)
    if concurrent:
        ev = torch.cuda.Event(); ev.record(main)
        with torch.cuda.stream(side):
            side.wait_event(ev)
            ssq = eng.perm_phase(k, inds=d_perm, M=Md, prepared=prep)
            done = torch.cuda.Event(); done.record(side)
    res = eng.boot_phase(k, inds=d_boot, M=Md, ref=ref, Xm=Xm, overlap_tail=True)
    if not concurrent:
        ssq = eng.perm_phase(k, inds=d_perm, M=Md, prepared=prep)
    with eng.tail_stream():
        out = eng.boot_finalize(res["S12"][0], res["S12"][1], phi - plo, num=ref)
    eng.join()
    if concurrent:
        main.wait_event(done)
    return out, ssq


for mode in (False, True, False, True):
    for _ in range(10):
        step(mode)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(mode)
    torch.cuda.synchronize()
    print(f"N={N} concurrent={mode}: {1e3 * (time.perf_counter() - t0) / steps:.4f} ms per step", flush=True)
