"""cProfile of a whole plspy_amd.PLS(...) call at config 2 (60 x 200 000, 1000 + 1000)."""
import cProfile, pstats, io, sys, time
import numpy as np
sys.path.insert(0, ".")
import plspy_amd
rs = np.random.RandomState(0)
X = rs.randn(60, 200_000)
def run():
    np.random.seed(1)
    return plspy_amd.PLS(X, (10, 10), 3, num_perm=1000, num_boot=1000, pls_method="mct")
run()
t0 = time.perf_counter(); run(); print("wall", time.perf_counter() - t0)
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("cumulative").print_stats(30); print(st.getvalue()[:5000])
