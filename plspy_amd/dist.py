"""Sharding of resample ids over the GPUs of one node (one process per GPU).

Resamples are independent, so the only exchange is one collective per phase
(RCCL over xGMI; gloo in the CPU tests): every rank contributes one packed
fp64 buffer -- its per-resample results (s_hat^2, Tdistrib numerators) and,
for the bootstrap, its shifted moment sums -- and receives everybody's.
Per-resample rows are concatenated in rank order; moment sums are added in
rank order, so every rank ends with bit-identical results.

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
    """One all_gather per phase.

    per_resample: list of tensors whose dim 0 is this rank's resample block.
    summed:       list of tensors to be added over ranks (moment sums).
    Returns (list of full per-resample tensors, list of rank-ordered sums)."""
    rank, n = world()
    if n == 1:
        return per_resample, summed
    out_dev = per_resample[0].device if per_resample else summed[0].device
    # RCCL moves device buffers over xGMI directly; any other backend (gloo in
    # the CPU tests) is staged through host memory
    dev = out_dev if td.get_backend() == "nccl" else torch.device("cpu")
    bounds = [shard_bounds(R, r, n) for r in range(n)]
    maxrows = max(hi - lo for lo, hi in bounds)
    row_elems = [int(np.prod(t.shape[1:])) for t in per_resample]
    sum_elems = [t.numel() for t in summed]
    width = maxrows * sum(row_elems) + sum(sum_elems)
    send = torch.zeros(width, dtype=torch.float64, device=dev)
    off = 0
    for t, re in zip(per_resample, row_elems):
        send[off:off + t.numel()] = t.reshape(-1).to(dev)
        off += maxrows * re
    for t, se in zip(summed, sum_elems):
        send[off:off + se] = t.reshape(-1).to(dev)
        off += se
    recv = [torch.empty_like(send) for _ in range(n)]
    td.all_gather(recv, send)
    full = []
    off = 0
    for t, re in zip(per_resample, row_elems):
        parts = [recv[r][off:off + (hi - lo) * re].reshape((hi - lo,) + tuple(t.shape[1:]))
                 for r, (lo, hi) in enumerate(bounds)]
        full.append(torch.cat(parts, dim=0).to(out_dev))
        off += maxrows * re
    sums = []
    for t, se in zip(summed, sum_elems):
        acc = recv[0][off:off + se].clone()
        for r in range(1, n):
            acc += recv[r][off:off + se]
        sums.append(acc.reshape(t.shape).to(out_dev))
        off += se
    return full, sums
