import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import test_gpu_stress as T
from plspy_amd.engine import ProjectionEngine
bad = 0
for seed in range(100, 108):
    rs = np.random.RandomState(seed)
    for trial in range(20):
        n = int(rs.randint(5, 200)); p = int(rs.choice([1, 15, 64, 65, 129, 300, 1000]))
        nz = int(rs.randint(4, 256)); ncell = int(rs.randint(1, min(nz, 20)))
        k = int(rs.choice([1, 5, 16, 17, 33, 48, 70, 100])); items = int(rs.randint(1, 12))
        cell_lo = T._cells(rs, nz, ncell); zflags = rs.randint(0, 2, size=ncell)
        X = rs.randn(n, p) * 2 + rs.randn(1, p) * 10
        src = rs.randint(0, n, size=(items, nz)).astype(np.int32)
        rows = rs.randn(items, k, nz); ref = rs.randn(p, k)
        eng = ProjectionEngine(X)
        S1 = torch.zeros((p, k), dtype=torch.float64, device=eng.device); S2 = torch.zeros_like(S1)
        try:
            vst, rowsq = eng.item_fused(src, cell_lo, zflags, rows, ref=ref, S1=S1, S2=S2, want_vst=True, want_rowsq=True)
        except Exception as e:
            print("unsupported", n, nz, k, ncell, str(e)[:60]); continue
        Z = T._zscore_items(X, src, cell_lo, zflags)
        want = np.einsum("bji,biv->bjv", rows, Z); scale = max(np.abs(want).max(), 1e-300)
        e1 = np.abs(vst.cpu().numpy() - want).max() / scale
        d = np.transpose(want, (0, 2, 1)) - ref
        e2 = np.abs(S2.cpu().numpy() - (d ** 2).sum(0)).max() / (scale + 1) ** 2
        e3 = np.abs(rowsq.cpu().numpy() - (want ** 2).sum(-1)).max() / scale ** 2
        if max(e1, e2, e3) > 1e-9:
            bad += 1; print("MISMATCH", seed, trial, n, p, nz, ncell, k, items, e1, e2, e3)
print("done, mismatches:", bad)
