"""Sharding of resample ids over the GPUs of one node (one process per GPU).

Resamples are independent, so the only exchange is one collective per phase
(RCCL over xGMI; gloo in the CPU tests): one all_gather of the per-resample
results (s_hat^2, Tdistrib numerators, packed; rows concatenated in rank order)
and, for the bootstrap, one all_reduce of the shifted moment sums.  Every rank ends
with bit-identical results.

Index vectors are drawn once, on rank 0, in the reference's RNG order and
broadcast, so results do not depend on the number of GPUs."""
import numpy as np
import torch
import torch.distributed as td


def world():
    if td.is_available() and td.is_initialized():
        return td.get_rank(), td.get_world_size()
    return 0, 1


def shard_bounds(R, rank, nranks):
    """Contiguous block of resample ids owned by ``rank``."""
    base, extra = divmod(R, nranks)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def broadcast_indices(inds, device=None):
    """Rank 0's (R x n) int32 index table, on every rank (as NumPy)."""
    rank, n = world()
    if n == 1:
        return inds
    backend = td.get_backend()
    dev = device if (backend == "nccl" and device is not None) else "cpu"
    shape = torch.tensor(list(inds.shape) if rank == 0 else [0, 0], dtype=torch.int64, device=dev)
    td.broadcast(shape, src=0)
    buf = (torch.as_tensor(np.ascontiguousarray(inds), dtype=torch.int32, device=dev) if rank == 0
           else torch.empty(tuple(int(x) for x in shape.tolist()), dtype=torch.int32, device=dev))
    td.broadcast(buf, src=0)
    return buf.cpu().numpy()


def exchange(per_resample, summed, R):
    """The collectives of one phase: ONE all_gather of the ranks' per-resample rows (all
    tensors packed into one buffer; small) and ONE all_reduce of the moment sums (2 p k
    doubles; an all_gather of those would move world_size times the data over xGMI).

    per_resample: list of tensors whose dim 0 is this rank's resample block.
    summed:       list of tensors to be added over ranks (moment sums).  A single contiguous
                  tensor that already lives where the backend reduces (the engine hands S1 / S2
                  over as one (2, p, k) block) is reduced IN PLACE -- no staging copy.
    Returns (list of full per-resample tensors, list of summed tensors); every rank ends with
    bit-identical results (RCCL / gloo reduce each element along one path and broadcast it).
    The sums depend on the number of ranks at rounding level (each rank adds its own block
    first); the per-resample rows do not."""
    rank, n = world()
    if n == 1:
        return per_resample, summed
    out_dev = per_resample[0].device if per_resample else summed[0].device
    # RCCL moves device buffers over xGMI directly; any other backend (gloo in
    # the CPU tests) is staged through host memory
    dev = out_dev if td.get_backend() == "nccl" else torch.device("cpu")
    full = []
    if per_resample:
        bounds = [shard_bounds(R, r, n) for r in range(n)]
        maxrows = max(hi - lo for lo, hi in bounds)
        even = all(hi - lo == maxrows for lo, hi in bounds)
        row_elems = [int(np.prod(t.shape[1:])) for t in per_resample]
        send = (torch.empty if even else torch.zeros)(maxrows * sum(row_elems), dtype=torch.float64, device=dev)
        off = 0
        for t, re in zip(per_resample, row_elems):
            send[off:off + t.numel()].copy_(t.reshape(-1))
            off += maxrows * re
        recv = torch.empty(n * send.numel(), dtype=torch.float64, device=dev)
        td.all_gather_into_tensor(recv, send)
        recv = recv.view(n, -1)
        off = 0
        for t, re in zip(per_resample, row_elems):
            block = recv[:, off:off + maxrows * re].reshape((n, maxrows) + tuple(t.shape[1:]))
            if even:
                got = block.reshape((n * maxrows,) + tuple(t.shape[1:]))
            else:
                got = torch.cat([block[r, :hi - lo] for r, (lo, hi) in enumerate(bounds)], dim=0)
            full.append(got.to(out_dev))
            off += maxrows * re
    sums = []
    if summed:
        if len(summed) == 1 and summed[0].device == dev and summed[0].is_contiguous():
            td.all_reduce(summed[0].view(-1), op=td.ReduceOp.SUM)
            sums = [summed[0]]
        else:
            flat = torch.cat([t.reshape(-1).to(dev) for t in summed])
            td.all_reduce(flat, op=td.ReduceOp.SUM)
            off = 0
            for t in summed:
                sums.append(flat[off:off + t.numel()].reshape(t.shape).to(out_dev))
                off += t.numel()
    return full, sums
