"""H2D upload of a 60 x 200 000 fp64 host array (pageable): one copy against row slices copied by several
threads (each thread's pageable copy stages its own slice; the GIL is released inside the copy)."""
import sys, threading, time
import numpy as np, torch
n, p = 60, 200_000
X = np.random.RandomState(0).randn(n, p)
dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
def single():
    return torch.from_numpy(X).to(dev)
def sliced(nth):
    out = torch.empty((n, p), dtype=torch.float64, device=dev)
    src = torch.from_numpy(X)
    streams = [torch.cuda.Stream(device=dev) for _ in range(nth)]
    def work(i):
        lo, hi = i * n // nth, (i + 1) * n // nth
        with torch.cuda.stream(streams[i]):
            out[lo:hi].copy_(src[lo:hi])
    ths = [threading.Thread(target=work, args=(i,)) for i in range(nth)]
    for t in ths: t.start()
    for t in ths: t.join()
    for s in streams: torch.cuda.current_stream().wait_stream(s)
    return out
for name, fn in [("single", single), ("2 threads", lambda: sliced(2)), ("3 threads", lambda: sliced(3)),
                 ("4 threads", lambda: sliced(4)), ("6 threads", lambda: sliced(6))]:
    for _ in range(2): fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(6):
        X[0, 0] += 1.0
        torch.cuda.synchronize(); t0 = time.perf_counter(); o = fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    assert float(o[0, 0]) == X[0, 0]
    print(f"{name:10s} {1e3 * min(ts):.2f} ms min, {1e3 * np.median(ts):.2f} ms median = {n * p * 8 / np.median(ts) / 1e9:.1f} GB/s")
