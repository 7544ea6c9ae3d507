"""Follow-up to cfg3_perm_bimodal.py: what makes the FIRST projection launch of a later run slow?  Variants
(argv[1]): 'sync' -- drain the device at the entry of perm_phase; 'sleep' -- and leave it idle for 50 ms;
'warm' -- run a 20 ms dummy projection right before; 'keepz' -- reuse one z-scored engine (no K3 + no new
engine per run)."""
import ctypes, sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
import plspy_amd
from plspy_amd import _lib
from plspy_amd import bootstrap_permutation as bp
from plspy_amd.bootstrap_permutation import ResampleTest
from plspy_amd.engine import ProjectionEngine

mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
lib = _lib.load()
X = np.random.RandomState(0).randn(120, 200_000)
Y = np.random.RandomState(1).randn(120, 8)
co = np.array([[20] * 3, [20] * 3])
np.random.seed(1234)
res = plspy_amd.PLS(X, [20, 20], 3, Y=Y, num_perm=0, num_boot=0, pls_method="rb")
U, s, V = res.V, res.s.copy(), res.U
eng = ProjectionEngine(X)
orig = ProjectionEngine.perm_phase


def patched(self, *a, **kw):
    if mode in ("sync", "sleep"):
        torch.cuda.synchronize()
    if mode == "sleep":
        time.sleep(0.05)
    return orig(self, *a, **kw)


ProjectionEngine.perm_phase = patched
for run in range(5):
    lib.plsr_timing_enable(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ResampleTest._create("rb", X, Y, U, s.copy(), V, co, None, nperm=2000, nboot=0, engine=eng)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    lib.plsr_timing_enable(0)
    ms = (ctypes.c_double * 64)()
    kind = (ctypes.c_int32 * 64)()
    nt = lib.plsr_timing_collect(ms, kind, 64)
    d = [round(ms[i], 2) for i in range(nt)]
    print(f"{mode} run {run}: wall {wall * 1e3:7.1f} ms, launches {d}, reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB, "
          f"allocated {torch.cuda.memory_allocated() / 2**30:.2f} GiB, gc counts {__import__('gc').get_count()}", flush=True)
    if mode == "gc":
        __import__("gc").collect()
