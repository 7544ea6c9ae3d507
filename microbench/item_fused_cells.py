"""Timing of K4f (plsr_item_fused) at config-3 shape as a function of the cell
structure: 6 cells of 20 rows (the real layout) against one cell of 120 rows
and 2 cells of 60 -- isolates the per-cell overhead of the kernel."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from plspy_amd.engine import ProjectionEngine

n, p, k, items = 120, 200_000, 48, 31
rs = np.random.RandomState(0)
X = rs.randn(n, p)
eng = ProjectionEngine(X)
src = rs.randint(0, n, size=(items, n)).astype(np.int32)
rows = rs.randn(items, k, n)
ref = rs.randn(p, k)
for cells in [(20,) * 6, (60, 60), (120,), (8,) * 15]:
    lo = np.concatenate(([0], np.cumsum(cells)))
    z = np.ones(len(cells), dtype=np.int32)
    S1 = torch.zeros((p, k), dtype=torch.float64, device=eng.device)
    S2 = torch.zeros_like(S1)
    d_src, d_rows, d_ref = eng.dev(src, torch.int32), eng.dev(rows), eng.dev(ref)
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.item_fused(d_src, lo, z, d_rows, ref=d_ref, S1=S1, S2=S2, want_vst=True, want_rowsq=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"cells {len(cells)} x {cells[0]}: {dt * 1e6 / items:.1f} us per item "
          f"({2 * n * k * p * items / dt / 1e12:.1f} TFLOP/s incl. stats + meta kernels)")

# which outputs cost what (one cell of 120 rows)
lo, z = np.array([0, 120]), np.ones(1, dtype=np.int32)
for name, kw in [("moments+vst+rowsq", dict(mom=True, want_vst=True, want_rowsq=True)),
                 ("vst only", dict(mom=False, want_vst=True, want_rowsq=False)),
                 ("rowsq only", dict(mom=False, want_vst=False, want_rowsq=True)),
                 ("moments only", dict(mom=True, want_vst=False, want_rowsq=False))]:
    mom = kw.pop("mom")
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.item_fused(d_src, lo, z, d_rows, ref=d_ref, S1=S1 if mom else None, S2=S2 if mom else None, **kw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"{name}: {dt * 1e6 / items:.1f} us per item")
