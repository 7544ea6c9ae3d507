"""Timing of K4a (plsr_item_agg) at config-3 shape, through the library named by PLSR_LIB
(ablation builds: hipcc -DAGG_ABLATE=<mask>).  Prints ms per launch and us per item."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from plspy_amd.engine import ProjectionEngine
n, p, k, items = 120, 200_000, 48, int(sys.argv[1]) if len(sys.argv) > 1 else 128
rs = np.random.RandomState(0)
eng = ProjectionEngine(rs.randn(n, p))
lo = np.arange(0, 121, 20)
src = np.concatenate([rs.randint(a, a + 20, size=(items, 20)) for a in lo[:-1]], axis=1).astype(np.int32)
d_src = eng.dev(src, torch.int32)
d_rows = eng.dev(rs.randn(items, k, n))
d_ref = eng.dev(rs.randn(p, k))
S1 = torch.zeros((p, k), dtype=torch.float64, device=eng.device)
S2 = torch.zeros_like(S1)
rng = eng.source_ranges(src, lo)
kw = dict(ref=d_ref, S1=S1, S2=S2, want_vst=True, want_rowsq=False, src_ranges=rng)
for _ in range(3):
    eng.item_fused(d_src, lo, np.ones(6, dtype=np.int32), d_rows, **kw)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 5
e0.record()
for _ in range(reps):
    eng.item_fused(d_src, lo, np.ones(6, dtype=np.int32), d_rows, **kw)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"{eng.last_item_kernel} items={items}: {ms:.3f} ms per call, {1e3 * ms / items:.1f} us per item "
      f"(whole call: meta + kernel + merge)")
