"""cProfile of the mb split-half (config 4 shape): where the host time goes."""
import cProfile, pstats, sys, io, time
import numpy as np
sys.path.insert(0, ".")
import torch
from plspy_amd import split_half_resampling as sh
from plspy_amd.engine import ProjectionEngine
X = np.random.RandomState(0).randn(120, 200_000)
Y = np.random.RandomState(1).randn(120, 8)
co = np.array([[20] * 3, [20] * 3])
eng = ProjectionEngine(X)
kw = dict(mctype=0, bscan=[1, 2], engine=eng)
np.random.seed(1)
S = 1000
for name, fn in (("test_train", lambda: sh.split_half_test_train("mb", X, Y, co, S, **kw)),
                 ("split_half", lambda: sh.split_half("mb", X, Y, co, S, lv=2, CI=0.95, **kw))):
    fn()
    torch.cuda.synchronize()
    pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable(); fn(); torch.cuda.synchronize(); pr.disable()
    print(name, "wall", time.perf_counter() - t0)
    st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(14); print(st.getvalue()[-2600:])
