"""The rb bootstrap pipeline at config-3 shape with the host's per-batch work switched off / replaced,
to find what makes one K4 launch in a dozen take 20-40 ms.  argv[1]: full | nohost | sleep | nocopy"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from plspy_amd.engine import ProjectionEngine
mode = sys.argv[1] if len(sys.argv) > 1 else "full"
n, p, b, ncell = 120, 200_000, 8, 6
k = ncell * b
rs = np.random.RandomState(0)
X = rs.randn(n, p)
eng = ProjectionEngine(X)
bounds = np.arange(0, 121, 20)
R = 2000
src = np.concatenate([rs.randint(a, a + 20, size=(R, 20)) for a in bounds[:-1]], axis=1).astype(np.int32)
Y = rs.randn(n, b)
U = np.linalg.qr(rs.randn(k, k))[0]
from plspy_amd import class_functions as cf
log = []
orig = eng.item_beh
def item_beh(*a, **kw):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); out = orig(*a, **kw); e1.record(); log.append((e0, e1)); return out
eng.item_beh = item_beh
def on_batch(a, z, zt, nsq):
    if mode == "full":
        Lt = np.take_along_axis(zt, src[a:z][:, None, :].astype(np.int64), axis=2)
        cf.lvcorr_from_latents(Lt, cf.zscore_cells(Y[src[a:z]], bounds), bounds)
    elif mode == "sleep":
        time.sleep(0.006)
if mode == "nocopy":
    on_batch = None
for rep in range(2):
    log.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.boot_items(src, bounds, np.ones(6), k, None, ref=rs.randn(p, k), on_batch=on_batch,
                   beh=(lambda a, z: cf.zscore_cells(Y[src[a:z]], bounds), U))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
t = np.array([a.elapsed_time(b) for a, b in log])
print(mode, f"wall {dt:.3f} s; K4 per batch median {np.median(t):.2f} ms, slow launches (ms):", np.round(t[t > 1.5 * np.median(t)], 1).tolist())
