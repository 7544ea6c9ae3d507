"""Where the observed phase of PLS(..., pls_method='rb') at config 3 spends its time (each statement of
pls_classes._BehaviourPLS.__init__ up to the resampling tests, synchronised and timed)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from plspy_amd import class_functions as cf
from plspy_amd.engine import ProjectionEngine

rs = np.random.RandomState(0)
groups, nc, p, nbeh = (20, 20), 3, 200_000, 8
co = np.array([[g] * nc for g in groups]); n = int(co.sum())
X = rs.randn(n, p); Y = rs.randn(n, nbeh)
bounds = cf.cell_bounds(co)
def stamp(label, t0):
    torch.cuda.synchronize(); t1 = time.perf_counter(); print(f"  {label:34s} {1e3 * (t1 - t0):8.2f} ms"); return t1
for rep in range(3):
    print("rep", rep)
    torch.cuda.synchronize(); t = time.perf_counter(); T0 = t
    engine = ProjectionEngine(X); t = stamp("upload X", t)
    Xz = engine.gather_zscore(np.arange(n), bounds, np.ones(len(bounds) - 1))[0]; t = stamp("gather_zscore", t)
    eng_z = ProjectionEngine(Xz, device=engine.device, work_limit=engine.work_limit); t = stamp("engine on Xz", t)
    A = cf.corr_operator(cf.zscore_cells(np.asarray(Y, dtype=float), bounds), bounds); t = stamp("corr_operator (host)", t)
    Rd = eng_z.apply_operator(A); t = stamp("apply_operator", t)
    R = Rd.cpu().numpy(); t = stamp("R .cpu()", t)
    U, s, V = eng_z.thin_svd(A); t = stamp("thin_svd (incl. downloads)", t)
    XL = engine.latents(V); t = stamp("latents(V) (upload V, K5, download)", t)
    YL = cf.compute_Y_latents(Y, U, co); t = stamp("Y latents (host)", t)
    lv = cf.compute_corr_small(XL, Y, co); t = stamp("lvcorrs (host)", t)
    print(f"  total {1e3 * (t - T0):.2f} ms")
