"""Does a pageable upload on a side stream block the host while a kernel runs on the main stream?
Times the upload call (host side) of a 5.4 MB / 56 KB array while a ~10 ms kernel is in flight, for
pageable and page-locked sources."""
import time
import numpy as np
import torch
side = torch.cuda.Stream()
big = torch.zeros(256 << 20, dtype=torch.float64, device="cuda")     # 2 GiB: add_ takes a few ms
def busy(n=3):
    for _ in range(n):
        big.add_(1.0)
torch.cuda.synchronize()
t0 = time.perf_counter(); busy(); torch.cuda.synchronize(); print("busy kernel block: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
pin = torch.empty(8 << 20, dtype=torch.uint8).pin_memory()
for nbytes in (56_000, 5_400_000):
    for mode in ("pageable", "pinned"):
        ts = []
        for rep in range(8):
            a = np.random.rand(nbytes // 8)
            src = torch.from_numpy(a)
            torch.cuda.synchronize()
            busy()
            t0 = time.perf_counter()
            with torch.cuda.stream(side):
                if mode == "pageable":
                    d = src.to("cuda")
                else:
                    st = pin[:nbytes // 8 * 8].view(torch.float64)
                    st.copy_(src)
                    d = st.to("cuda", non_blocking=True)
            ts.append((time.perf_counter() - t0) * 1e3)
            torch.cuda.synchronize()
        print(f"{nbytes/1e6:6.3f} MB {mode:9s} under load: " + " ".join(f"{t:6.2f}" for t in ts) + " ms", flush=True)
