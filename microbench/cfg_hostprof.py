"""cProfile of one warm resampling call: `cfg_hostprof.py 3 perm|boot` (rb, config 3) or `6 perm|boot` (mb,
config 6) -- where the host spends what the device's kernels do not cover."""
import cProfile, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import plspy_amd
from plspy_amd.bootstrap_permutation import ResampleTest
from plspy_amd.engine import ProjectionEngine

cfg, phase = int(sys.argv[1]), sys.argv[2]
rs = np.random.RandomState(0)
X = rs.randn(120, 200_000); Y = rs.randn(120, 8)
co = np.array([[20] * 3, [20] * 3])
R = int(sys.argv[3]) if len(sys.argv) > 3 else (2000 if cfg == 3 else 500)
np.random.seed(1)
if cfg == 3:
    res = plspy_amd.PLS(X, [20, 20], 3, Y=Y, num_perm=0, num_boot=0, pls_method="rb")
    kw = dict(lvcorrs_orig=res.lvcorrs)
    alg, mct = "rb", None
else:
    res = plspy_amd.PLS(X, [20, 20], 3, Y=Y, num_perm=0, num_boot=0, pls_method="mb", bscan=[1, 2])
    kw = dict(lvcorrs_orig=res.lvcorrs, Tvsc_orig=np.zeros((6, len(res.s))), bscan=[1, 2], Xbscan=res.Xbscan, Ybscan=res.Ybscan)
    alg, mct = "mb", 0
U, s, V = res.V, res.s.copy(), res.U
eng = ProjectionEngine(X)
def run():
    return ResampleTest._create(alg, X, Y, U, s.copy(), V, co, mct, nperm=R if phase == "perm" else 0,
                                nboot=R if phase == "boot" else 0, engine=eng, **kw)
run()
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); run(); torch.cuda.synchronize()
    print(f"wall {time.perf_counter() - t0:.4f} s -> {R / (time.perf_counter() - t0):.0f} /s")
pr = cProfile.Profile(); pr.enable(); run(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr, stream=sys.stdout).sort_stats("tottime").print_stats(22)
