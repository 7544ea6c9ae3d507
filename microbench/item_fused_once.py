"""One plsr_item_fused call at config-3 shape (for rocprofv3 --pmc passes)."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from plspy_amd.engine import ProjectionEngine
n, p, k, items = 120, 200_000, 48, 31
rs = np.random.RandomState(0)
eng = ProjectionEngine(rs.randn(n, p))
src = rs.randint(0, n, size=(items, n)).astype(np.int32)
rows = rs.randn(items, k, n)
cells = (20,) * 6 if len(sys.argv) < 2 else tuple(int(x) for x in sys.argv[1].split(","))
lo = np.concatenate(([0], np.cumsum(cells)))
S1 = torch.zeros((p, k), dtype=torch.float64, device=eng.device)
S2 = torch.zeros_like(S1)
for _ in range(2):
    eng.item_fused(src, lo, np.ones(len(cells), dtype=np.int32), rows, ref=rs.randn(p, k), S1=S1, S2=S2,
                   want_vst=True, want_rowsq=True)
torch.cuda.synchronize()
