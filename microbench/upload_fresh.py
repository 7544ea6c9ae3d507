"""Host time of pageable uploads of FRESH host buffers (as ProjectionEngine.dev sees them) by size."""
import time
import numpy as np
import torch
side = torch.cuda.Stream()
torch.zeros(1, device="cuda")
for mb in (0.003, 0.24, 0.9, 1.2, 4.8, 9.6, 19.2, 31, 40, 96):
    n = int(mb * 1e6 / 8)
    ts = []
    for rep in range(5):
        a = np.random.rand(n)            # fresh pages every time
        src = torch.from_numpy(a)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(side):
            dst = src.to("cuda")
        ts.append((time.perf_counter() - t0) * 1e3)
        torch.cuda.synchronize()
        del a, src
    print(f"{mb:7.3f} MB fresh: " + " ".join(f"{t:7.3f}" for t in ts) + " ms", flush=True)
# persistent pinned staging buffer
pin = torch.empty(int(20e6 / 8), dtype=torch.float64).pin_memory()
for mb in (0.24, 9.6):
    n = int(mb * 1e6 / 8)
    ts = []
    for rep in range(5):
        a = np.random.rand(n)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pin[:n].copy_(torch.from_numpy(a))
        with torch.cuda.stream(side):
            dst = pin[:n].to("cuda", non_blocking=True)
        ts.append((time.perf_counter() - t0) * 1e3)
        torch.cuda.synchronize()
    print(f"{mb:7.3f} MB via persistent pinned buffer: " + " ".join(f"{t:7.3f}" for t in ts) + " ms", flush=True)
