"""cProfile (by own time) of a warm plspy_amd.PLS(...) call at config 2 (60 x 200 000, 1000 + 1000),
with the number of fresh device allocations (hipMalloc) the call caused."""
import cProfile, pstats, io, sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
import plspy_amd
rs = np.random.RandomState(0)
X = rs.randn(60, 200_000)
def run():
    np.random.seed(1)
    return plspy_amd.PLS(X, (10, 10), 3, num_perm=1000, num_boot=1000, pls_method="mct")
run(); run()
for _ in range(3):
    n0 = torch.cuda.memory_stats()["num_device_alloc"]
    torch.cuda.synchronize(); t0 = time.perf_counter(); run(); torch.cuda.synchronize()
    print("wall ms", (time.perf_counter() - t0) * 1e3, "device allocations", torch.cuda.memory_stats()["num_device_alloc"] - n0)
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(22); print(st.getvalue()[:6000])
