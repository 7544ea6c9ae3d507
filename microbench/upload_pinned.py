"""How should the host matrix X reach HBM?  Times, for 96 MB (config 2) and 960 MB
(config 5): a pageable copy (what Tensor.to does), hipHostRegister of the caller's
buffer in place + async copy + unregister, and a chunked register/copy pipeline."""
import time

import numpy as np
import torch

rt = torch.cuda.cudart()


def t(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


for n, p in ((60, 200_000), (240, 500_000)):
    X = np.random.RandomState(0).randn(n, p)
    src = torch.from_numpy(X)
    dst = torch.empty((n, p), dtype=torch.float64, device="cuda")

    def pageable():
        dst.copy_(src)

    def registered():
        err = rt.cudaHostRegister(src.data_ptr(), src.numel() * 8, 0)
        assert int(err) == 0, err
        try:
            assert src.is_pinned()
            dst.copy_(src, non_blocking=True)
            torch.cuda.synchronize()
        finally:
            rt.cudaHostUnregister(src.data_ptr())

    def chunked(chunk_rows=max(1, n // 8)):
        # register chunk i+1 while chunk i is in flight
        regs = []
        for lo in range(0, n, chunk_rows):
            hi = min(n, lo + chunk_rows)
            s = src[lo:hi]
            err = rt.cudaHostRegister(s.data_ptr(), s.numel() * 8, 0)
            assert int(err) == 0, err
            regs.append(s.data_ptr())
            dst[lo:hi].copy_(s, non_blocking=True)
        torch.cuda.synchronize()
        for ptr in regs:
            rt.cudaHostUnregister(ptr)

    print(f"{n}x{p} ({X.nbytes / 1e6:.0f} MB): pageable {t(pageable):.1f} ms, registered {t(registered):.1f} ms, "
          f"chunked {t(chunked):.1f} ms", flush=True)
    got = dst.cpu().numpy()
    assert np.array_equal(got, X)
