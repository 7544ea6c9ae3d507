"""Timing of K5 (plsr_latent) at config-3 shape: ms per launch of `items` items."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from plspy_amd import _lib
from plspy_amd.engine import ProjectionEngine, _ptr, _stream
n, p, k, items = 120, 200_000, 48, int(sys.argv[1]) if len(sys.argv) > 1 else 125
rs = np.random.RandomState(0)
eng = ProjectionEngine(rs.randn(n, p))
vst = torch.randn((items, k, p), dtype=torch.float64, device=eng.device)
need = eng.lib.plsr_latent_workspace_bytes(n, k, items, p)
work = torch.empty(need, dtype=torch.uint8, device=eng.device)
Zt = torch.empty((items, k, n), dtype=torch.float64, device=eng.device)
nsq = torch.empty((items, k), dtype=torch.float64, device=eng.device)
def run():
    if "xt" in sys.argv:
        eng.latent_batch(vst, n, Zt, nsq); return
    _lib.check(eng.lib.plsr_latent(_ptr(eng.X), eng.X.stride(0), p, n, _ptr(vst), p, items, k, _ptr(Zt), _ptr(nsq) if "nonsq" not in sys.argv else _ptr(None),
                                   _ptr(work), need, _stream()), "plsr_latent")
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
want = torch.einsum("jv,iv->ji", vst[3], eng.X)
err = (Zt[3] - want).abs().max().item() / want.abs().max().item()
print(f"latent items={items}: {ms:.3f} ms per call, {1e3 * ms / items:.1f} us per item, rel err {err:.1e}")
