"""Per-launch durations of K4b (125 items, config-3 shape) over many launches, with and without
the VS^T stores: looking for the sporadic 20-40 ms launches."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from plspy_amd.engine import ProjectionEngine
n, p, b, ncell, items = 120, 200_000, 8, 6, 125
k = ncell * b
rs = np.random.RandomState(0)
eng = ProjectionEngine(rs.randn(n, p))
lo = np.arange(0, 121, 20)
src = np.concatenate([rs.randint(a, a + 20, size=(items, 20)) for a in lo[:-1]], axis=1).astype(np.int32)
Y = rs.randn(items, n, b)
Yz = np.concatenate([(Y[:, a:a + 20] - Y[:, a:a + 20].mean(1, keepdims=True)) / Y[:, a:a + 20].std(1, keepdims=True)
                     for a in lo[:-1]], axis=1)
d_src, d_Yz, d_U, d_ref = eng.dev(src, torch.int32), eng.dev(Yz), eng.dev(rs.randn(k, k)), eng.dev(rs.randn(p, k))
S1 = torch.zeros((p, k), dtype=torch.float64, device=eng.device); S2 = torch.zeros_like(S1)
rng = eng.source_ranges(src, lo)
for mode in sys.argv[1:] or ["vst", "novst"]:
    kw = dict(ref=d_ref, S1=S1, S2=S2, want_vst=(mode == "vst"), pool=True)
    evs = []
    for _ in range(80):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); eng.item_beh(d_src, lo, rng, d_Yz, d_U, **kw); e1.record(); evs.append((e0, e1))
    torch.cuda.synchronize()
    t = np.array([a.elapsed_time(b) for a, b in evs])
    print(mode, "median %.2f ms, max %.2f ms, launches over 1.5x median: %s" % (np.median(t), t.max(), np.flatnonzero(t > 1.5 * np.median(t)).tolist()),
          "their ms:", np.round(t[t > 1.5 * np.median(t)], 1).tolist())
