"""rb bootstrap at config 3: wall time of the phase against the span of its device work (first VS kernel .. last kernel)
and the host time before / after that span."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
from plspy_amd.bootstrap_permutation import ResampleTest
from plspy_amd.engine import ProjectionEngine
rs = np.random.RandomState(0)
co = np.array([[20] * 3, [20] * 3]); n = 120; p = 200_000; b = 8
X = rs.randn(n, p); Y = rs.randn(n, b)
eng = ProjectionEngine(X)
k = b * co.size
if len(sys.argv) > 1:                      # the observed decomposition, as bench_configs.py passes it
    import plspy_amd
    res = plspy_amd.PLS(X, [20, 20], 3, Y=Y, num_perm=0, num_boot=0, pls_method="rb")
    U, s, V = res.V, res.s.copy(), res.U
    print("observed: U", U.shape, U.flags["C_CONTIGUOUS"], "V", V.shape, V.flags["C_CONTIGUOUS"], V.dtype, V.strides)
else:
    U, _ = np.linalg.qr(rs.randn(k, k)); s = np.abs(rs.randn(k)) + 1; V = rs.randn(p, k)
np.random.seed(1)
marks = {}
orig = eng.item_beh
def hooked(*a, **kw):
    if "first" not in marks:
        marks["t_first_host"] = time.perf_counter()
        ev = torch.cuda.Event(enable_timing=True); ev.record(); marks["first"] = ev
    return orig(*a, **kw)
eng.item_beh = hooked
def run():
    marks.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = ResampleTest._create("rb", X, Y, U, s.copy(), V, co, 0, nperm=0, nboot=2000, lvcorrs_orig=np.zeros((k, k)), engine=eng)
    t1 = time.perf_counter()
    ev = torch.cuda.Event(enable_timing=True); ev.record(); torch.cuda.synchronize()
    return t1 - t0, marks["t_first_host"] - t0, marks["first"].elapsed_time(ev) * 1e-3
run()
for _ in range(3):
    wall, front, span = run()
    print(f"wall {wall*1e3:.1f} ms; host before the first VS kernel {front*1e3:.1f} ms; first VS kernel -> return {span*1e3:.1f} ms", flush=True)
