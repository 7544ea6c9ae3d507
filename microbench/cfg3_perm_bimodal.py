"""Config 3's permutation phase (rb, 2000 permutations of Y against the z-scored 120 x 200 000 X), several times
in one process: per run the wall time, the hipEvent durations of its projection launches (plsr_timing_*) and the
device clock the driver reports (pp_dpm_sclk) before and after -- what separates a slower kernel from a slower
host."""
import ctypes, glob, sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
import plspy_amd
from plspy_amd import _lib
from plspy_amd.bootstrap_permutation import ResampleTest
from plspy_amd.engine import ProjectionEngine


def sclk():
    out = []
    for f in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
        try:
            cur = [ln.split()[1] for ln in open(f) if "*" in ln]
            out.append(cur[0] if cur else "?")
        except OSError:
            pass
    return ",".join(out[:2]) or "n/a"


lib = _lib.load()
X = np.random.RandomState(0).randn(120, 200_000)
Y = np.random.RandomState(1).randn(120, 8)
co = np.array([[20] * 3, [20] * 3])
np.random.seed(1234)
res = plspy_amd.PLS(X, [20, 20], 3, Y=Y, num_perm=0, num_boot=0, pls_method="rb")
U, s, V = res.V, res.s.copy(), res.U
eng = ProjectionEngine(X)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
pause = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
for run in range(6):
    if pause:
        time.sleep(pause)
    c0 = sclk()
    lib.plsr_timing_enable(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ResampleTest._create("rb", X, Y, U, s.copy(), V, co, None, nperm=R, nboot=0, engine=eng)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    c1 = sclk()
    lib.plsr_timing_enable(0)
    ms = (ctypes.c_double * 64)()
    kind = (ctypes.c_int32 * 64)()
    nt = lib.plsr_timing_collect(ms, kind, 64)
    d = [round(ms[i], 2) for i in range(nt)]
    print(f"run {run}: wall {wall * 1e3:7.1f} ms, launches {d} = {sum(d):6.1f} ms, outside the launches {wall * 1e3 - sum(d):5.1f} ms, "
          f"sclk {c0} -> {c1}", flush=True)
