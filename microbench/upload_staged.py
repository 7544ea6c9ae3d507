import sys, time, os
sys.path.insert(0, ".")
import numpy as np, torch
from plspy_amd import engine as E
X = np.random.randn(16, 1000)
eng = E.ProjectionEngine(X)
for nbytes in (100_000, 1_000_000, 4_000_000, 30_000_000):
    a = np.random.randn(nbytes // 8)
    for mode in ("pageable", "staged"):
        E.PIN_UPLOAD_MIN, E.PIN_UPLOAD_MAX = ((64 << 10), (8 << 20)) if mode == "staged" else (1, 0)
        for _ in range(3):
            eng.dev(a)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            t = eng.dev(a)
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        print(f"{nbytes:>10} B {mode:9s}: host {t_host / 50 * 1e3:7.3f} ms per upload, with sync {(time.perf_counter() - t0) / 50 * 1e3:7.3f} ms")
