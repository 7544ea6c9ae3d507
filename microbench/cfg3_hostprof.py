"""cProfile of one warm rb bootstrap call at config 3 (2000 bootstraps): where the host spends the
part of the 0.26 s that the device (0.17 s of kernels) does not cover."""
import cProfile, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from plspy_amd.bootstrap_permutation import ResampleTest
from plspy_amd.engine import ProjectionEngine

rs = np.random.RandomState(0)
groups, nc, p, nbeh = (20, 20), 3, 200_000, 8
co = np.array([[g] * nc for g in groups])
n = int(co.sum())
X = rs.randn(n, p); Y = rs.randn(n, nbeh)
eng = ProjectionEngine(X)
k = nbeh * co.size
U, _ = np.linalg.qr(rs.randn(k, k)); s = np.abs(rs.randn(k)) + 1; V = rs.randn(p, k)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
def run():
    return ResampleTest._create("rb", X, Y, U, s.copy(), V, co, 0, nperm=0, nboot=R, lvcorrs_orig=np.zeros((k, k)), engine=eng)
np.random.seed(1); run()
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); run(); torch.cuda.synchronize()
    print(f"wall {time.perf_counter() - t0:.4f} s -> {R / (time.perf_counter() - t0):.0f} /s")
pr = cProfile.Profile(); pr.enable(); run(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr, stream=sys.stdout).sort_stats("tottime").print_stats(25)
pstats.Stats(pr, stream=sys.stdout).sort_stats("cumtime").print_stats(30)
