"""Timeline of the rb bootstrap at config 3: host time per boot_items phase and device time per batch
(torch events around every batch), to see whether the device or the host paces the pipeline."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from plspy_amd.bootstrap_permutation import ResampleTest
from plspy_amd.engine import ProjectionEngine
import plspy_amd.engine as E

rs = np.random.RandomState(0)
groups, nc, p, nbeh = (20, 20), 3, 200_000, 8
co = np.array([[g] * nc for g in groups])
n = int(co.sum())
X = rs.randn(n, p); Y = rs.randn(n, nbeh)
eng = ProjectionEngine(X)
k = nbeh * co.size
U, _ = np.linalg.qr(rs.randn(k, k)); s = np.abs(rs.randn(k)) + 1; V = rs.randn(p, k)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
log = []
orig_item_fused, orig_dev = eng.item_fused, eng.dev
def item_fused(*a, **kw):
    t0 = time.perf_counter(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); out = orig_item_fused(*a, **kw); e1.record()
    log.append(("item_fused", t0, time.perf_counter(), e0, e1)); return out
eng.item_fused = item_fused
orig_item_beh = eng.item_beh
def item_beh(*a, **kw):
    t0 = time.perf_counter(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); out = orig_item_beh(*a, **kw); e1.record()
    if out is not None: log.append(("item_beh", t0, time.perf_counter(), e0, e1))
    return out
eng.item_beh = item_beh
lat = eng.lib.plsr_latent
class L:
    def __getattr__(self, name): return getattr(eng_lib, name)
eng_lib = eng.lib
def run():
    return ResampleTest._create("rb", X, Y, U, s.copy(), V, co, 0, nperm=0, nboot=R, lvcorrs_orig=np.zeros((k, k)), engine=eng)
np.random.seed(1); run(); log.clear()
torch.cuda.synchronize(); T0 = time.perf_counter(); run(); torch.cuda.synchronize(); T1 = time.perf_counter()
print(f"wall {T1 - T0:.4f} s for {R} boots = {R / (T1 - T0):.0f} /s")
base = log[0][1]
prev_end = None
for name, t0, t1, e0, e1 in log:
    gap = "" if prev_end is None else f" device gap since previous batch's K4 end {prev_end.elapsed_time(e0):7.2f} ms"
    print(f"{name}: host enqueue at {1e3 * (t0 - base):7.2f} ms (took {1e3 * (t1 - t0):5.2f}), device {e0.elapsed_time(e1):6.2f} ms{gap}")
    prev_end = e1
