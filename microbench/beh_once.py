"""Timing of K4b (plsr_item_beh) at config-3 shape: ms per call of `items` items."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from plspy_amd.engine import ProjectionEngine
n, p, b, ncell, items = 120, 200_000, 8, 6, int(sys.argv[1]) if len(sys.argv) > 1 else 128
k = ncell * b
rs = np.random.RandomState(0)
eng = ProjectionEngine(rs.randn(n, p))
lo = np.arange(0, 121, 20)
src = np.concatenate([rs.randint(a, a + 20, size=(items, 20)) for a in lo[:-1]], axis=1).astype(np.int32)
Y = rs.randn(items, n, b)
Yz = np.concatenate([(Y[:, a:a + 20] - Y[:, a:a + 20].mean(1, keepdims=True)) / Y[:, a:a + 20].std(1, keepdims=True)
                     for a in lo[:-1]], axis=1)
d_src, d_Yz, d_U, d_ref = eng.dev(src, torch.int32), eng.dev(Yz), eng.dev(rs.randn(k, k)), eng.dev(rs.randn(p, k))
S1 = torch.zeros((p, k), dtype=torch.float64, device=eng.device)
S2 = torch.zeros_like(S1)
rng = eng.source_ranges(src, lo)
mode = sys.argv[2] if len(sys.argv) > 2 else 'full'
kw = dict(ref=d_ref, S1=S1, S2=S2, want_vst=True)
if mode == 'novst': kw['want_vst'] = False
if mode == 'nomom': kw.update(S1=None, S2=None, ref=None)
if mode == 'none': kw = dict(want_vst=False)
for _ in range(3):
    eng.item_beh(d_src, lo, rng, d_Yz, d_U, **kw)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    eng.item_beh(d_src, lo, rng, d_Yz, d_U, **kw)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"beh {mode} items={items}: {ms:.3f} ms per call, {1e3 * ms / items:.1f} us per item (whole call: meta + kernel + merge)")
