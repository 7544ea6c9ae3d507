"""Cost of first-use device allocations (hipMalloc through torch + first touch) by size."""
import time
import torch
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
for gib in (0.25, 1, 2, 4, 8, 16):
    n = int(gib * (1 << 30))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    t = torch.empty(n, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize(); t1 = time.perf_counter()
    t.zero_(); torch.cuda.synchronize(); t2 = time.perf_counter()
    t.zero_(); torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"{gib:6.2f} GiB: alloc {1e3*(t1-t0):8.1f} ms, first touch {1e3*(t2-t1):8.1f} ms, second touch {1e3*(t3-t2):8.1f} ms", flush=True)
    del t
    torch.cuda.empty_cache()
