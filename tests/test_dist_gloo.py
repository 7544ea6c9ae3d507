"""N > 1 path on CPU: two gloo processes shard a phase, exchange once and must
end with the single-process answer on every rank.  The per-rank "kernel" here
is the oracle's arithmetic (tests may use it); what is under test is the
product's sharding / broadcast / packed all_gather / rank-ordered merge."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as td
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_phase(X, W, U, ref, inds):
    """What a rank's GPU would return for its block of bootstrap resamples."""
    VS = np.stack([(W @ X[i]).T @ U for i in inds]) if len(inds) else np.zeros((0, X.shape[1], U.shape[1]))
    ssq = (VS ** 2).sum(axis=1)
    T = np.einsum("bvj,cv->bjc", VS, X[: U.shape[0]])
    d = VS - ref
    return (torch.from_numpy(ssq), torch.from_numpy(T),
            torch.from_numpy(d.sum(axis=0)), torch.from_numpy((d ** 2).sum(axis=0)))


def _worker(rank, world, port, R, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    td.init_process_group("gloo", rank=rank, world_size=world)
    from plspy_amd import dist, operators, resample
    co = np.array([[3, 3], [2, 2]])
    rs = np.random.RandomState(0)
    X = rs.randn(10, 37)
    W = operators.mean_centre_operator(co, 0)
    U = np.linalg.svd(W @ X, full_matrices=False)[0]
    ref = rs.randn(37, 4)
    # only rank 0 draws; its RNG state is what the reference's would be
    np.random.seed(99)
    inds = resample.bootstraps(co, R) if rank == 0 else None
    inds = dist.broadcast_indices(inds)
    lo, hi = dist.shard_bounds(R, rank, world)
    ssq, T, S1, S2 = _fake_phase(X, W, U, ref, inds[lo:hi])
    (ssq, T), (S1, S2) = dist.exchange([ssq, T], [S1, S2], R)
    q.put((rank, inds, ssq.numpy(), T.numpy(), S1.numpy(), S2.numpy()))
    td.barrier()
    td.destroy_process_group()


@pytest.mark.parametrize("R", [7, 2, 1])
def test_two_rank_exchange_matches_single_process(R):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, R, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0

    sys.path.insert(0, ROOT)
    from plspy_amd import operators, resample
    co = np.array([[3, 3], [2, 2]])
    rs = np.random.RandomState(0)
    X = rs.randn(10, 37)
    W = operators.mean_centre_operator(co, 0)
    U = np.linalg.svd(W @ X, full_matrices=False)[0]
    ref = rs.randn(37, 4)
    np.random.seed(99)
    inds = resample.bootstraps(co, R)
    ssq, T, S1, S2 = [t.numpy() for t in _fake_phase(X, W, U, ref, inds)]
    for rank, ginds, gssq, gT, gS1, gS2 in got:
        np.testing.assert_array_equal(ginds, inds)
        np.testing.assert_array_equal(gssq, ssq)          # per-resample rows: bit-identical
        np.testing.assert_array_equal(gT, T)
        np.testing.assert_allclose(gS1, S1, rtol=1e-13, atol=1e-13)   # sums: rank-ordered add
        np.testing.assert_allclose(gS2, S2, rtol=1e-13, atol=1e-13)
    # every rank ends with bit-identical results
    for a, b in zip(got[0][2:], got[1][2:]):
        np.testing.assert_array_equal(a, b)
