"""cProfile of the mb bootstrap (config 6 shape, 500 resamples): where the host time goes."""
import cProfile, pstats, sys, io, time
import numpy as np
sys.path.insert(0, ".")
import torch
import plspy_amd
from plspy_amd.bootstrap_permutation import ResampleTest
from plspy_amd.engine import ProjectionEngine
from oracle import plspy_oracle as orc
X = np.random.RandomState(0).randn(120, 200_000)
Y = np.random.RandomState(1).randn(120, 8)
co = np.array([[20] * 3, [20] * 3])
bscan = [1, 2]
np.random.seed(1234)
res = plspy_amd.PLS(X, [20, 20], 3, Y=Y, num_perm=0, num_boot=0, pls_method="mb", bscan=bscan)
U, s, V = res.V, res.s.copy(), res.U
eng = ProjectionEngine(X)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 500
def run():
    return ResampleTest._create("mb", X, Y, U, s.copy(), V, co, 0, nperm=0, nboot=R, bscan=bscan, Xbscan=res.Xbscan,
                                Ybscan=res.Ybscan, lvcorrs_orig=res.lvcorrs, Tvsc_orig=np.zeros((6, 38)), engine=eng)
run()
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); run(); torch.cuda.synchronize()
    print("wall", time.perf_counter() - t0)
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(22); print(st.getvalue()[:5000])
