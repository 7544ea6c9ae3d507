"""cProfile of the rb bootstrap (config 3 shape, 310 resamples): where the host time goes."""
import cProfile, pstats, sys, io
import numpy as np
sys.path.insert(0, ".")
import bench_configs as bc
from plspy_amd.bootstrap_permutation import ResampleTest
from plspy_amd.engine import ProjectionEngine
from plspy_amd import class_functions as cf

rs = np.random.RandomState(0)
groups, nc, p, nbeh = (20, 20), 3, 200_000, 8
co = np.array([[g] * nc for g in groups])
n = int(co.sum())
X = rs.randn(n, p)
Y = rs.randn(n, nbeh)
eng = ProjectionEngine(X)
R = cf.compute_corr_small(X[:, :2000], Y, co) if hasattr(cf, "compute_corr_small") else None
k = nbeh * co.size
U, _ = np.linalg.qr(rs.randn(k, k))
s = np.abs(rs.randn(k)) + 1
V = rs.randn(p, k)
np.random.seed(1)
import time
def run():
    return ResampleTest._create("rb", X, Y, U, s.copy(), V, co, 0, nperm=0, nboot=2000,
                                lvcorrs_orig=np.zeros((k, k)), engine=eng)
run()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
run()
pr.disable()
print("wall", time.perf_counter() - t0)
st = io.StringIO()
pstats.Stats(pr, stream=st).sort_stats("cumulative").print_stats(28)
print(st.getvalue()[:6000])
