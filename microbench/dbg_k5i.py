import sys
import numpy as np
sys.path.insert(0, ".")
import torch
from plspy_amd.engine import ProjectionEngine
for (n, k, p, items, m) in [(128, 16, 515, 4, 128), (128, 16, 512, 4, 128), (128, 16, 2048, 2, 128), (112, 16, 515, 4, 112), (128, 48, 515, 2, 96)]:
    rs = np.random.RandomState(1)
    X = rs.randn(n, p)
    eng = ProjectionEngine(X)
    vs = rs.randn(items, k, p)
    idx = np.stack([rs.permutation(n)[:m] for _ in range(items)]).astype(np.int32)
    d_vs, d_idx = eng.dev(vs), eng.dev(idx, torch.int32)
    want = np.einsum("bjv,biv->bji", vs, X[idx])
    L = torch.full((items, k, m), float("nan"), dtype=torch.float64, device=eng.device)
    eng.latent_batch_index(d_vs, n, idx, d_idx, L, None)
    got = L.cpu().numpy()
    err = np.abs(got - want)
    print((n, k, p, items, m), eng.last_latent_kernel, "max err", err.max())
    if err.max() > 1e-8:
        bad = err > 1e-8
        print(" bad per item", bad.reshape(items, -1).sum(1), "per lv row", bad.sum((0, 2)), )
        # which X rows (sorted list position) are wrong
        pos = np.argsort(np.argsort(idx, axis=1), axis=1)
        for b in range(min(items, 2)):
            badcols = np.flatnonzero(bad[b].any(0))
            print(" item", b, "bad sorted positions", sorted(set(pos[b][badcols] // 16)))
        # try: is got == want computed with a subset of voxels?
        for vcut in (480, 496, 504, 512):
            w2 = np.einsum("bjv,biv->bji", vs[:, :, :vcut], X[idx][:, :, :vcut])
            print("  voxels <", vcut, np.abs(got - w2).max())
