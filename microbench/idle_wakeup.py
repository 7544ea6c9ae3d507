"""Does the device (or the runtime) make the first operation after a short idle period slow?
After `idle` ms without GPU work: host time of a 3 KB pageable upload, and of a tiny kernel + sync."""
import time
import numpy as np
import torch
a = np.zeros((6, 60))
src = torch.from_numpy(a)
dst = torch.empty((6, 60), dtype=torch.float64, device="cuda")
big = torch.zeros(64 << 20, dtype=torch.float64, device="cuda")
side = torch.cuda.Stream()
for mode in ("upload", "kernel", "upload_side_stream", "upload_after_busy"):
    for idle in (0, 1, 2, 5, 10, 20, 50):
        ts = []
        for rep in range(6):
            big.add_(1.0)                       # ~0.2 ms of device work
            torch.cuda.synchronize()
            time.sleep(idle * 1e-3)
            if mode == "upload_after_busy":
                big.add_(1.0); big.add_(1.0); big.add_(1.0)
            t0 = time.perf_counter()
            if mode == "kernel":
                dst.add_(1.0); torch.cuda.synchronize()
            elif mode == "upload_side_stream":
                with torch.cuda.stream(side):
                    dst.copy_(src)
            else:
                dst.copy_(src)
            ts.append((time.perf_counter() - t0) * 1e3)
            torch.cuda.synchronize()
        print(f"{mode:22s} idle {idle:3d} ms: " + " ".join(f"{t:7.3f}" for t in ts), flush=True)
