"""Split-half reproducibility tests on the GPU -- drop-in for
plspy/core/split_half_resampling.py (split_half_test_train :23-401,
split_half :404-861).

Per split the reference gathers both halves of X, preprocesses them, runs a
LAPACK SVD on each (k x p) block and multiplies the singular vectors.  Here a
split is one *item* of the Gram kernel: the two halves' operators (rows of
``W_half @ P_half``) are stacked, the kernel returns

    G = [M1; M2] [M1; M2]^T      M_h = A_h X  (never stored)

and everything the reference derives from the SVDs follows from the k x k
blocks G11, G12, G22 and the Jacobi eigen-decompositions of G11 / G22:

    U_h, s_h^2            = eig(G_hh)
    V1^T M2^T U1          = S1^-1 U1^T G12 U1                 (:196)
    V1^T V2               = S1^-1 U1^T G12 U2 S2^-1           (:682)
    U1^T U2                                                    (:683)

Random draws follow the reference's np.random call order.  Signs of singular
vectors are arbitrary in the reference (LAPACK) and here (Jacobi); the summary
statistics use absolute values of the diagonal and are sign-free.  Latent
variables that are null by construction (rank-deficient mean-centring) have
arbitrary vectors in the reference; here their singular value is deflated to 0
and their entries are 0."""
import numpy as np

from . import dist, exceptions, operators, resample
from .engine import ProjectionEngine


def _get_cond_order(X_shape, groups_tuple, num_conditions):
    """split_half_resampling.py:5-21."""
    if sum(groups_tuple) * num_conditions != X_shape[0]:
        raise exceptions.InputMatrixDimensionMismatchError(
            "Derived condition ordering not compatible with input matrix"
            "X's row count. Please specify a custom cond_order field.")
    return np.array([np.array([i] * num_conditions) for i in groups_tuple])


def _draw_splits(cond_order, num_split, n, task):
    """Index vectors of every real and null split, in the reference's RNG order
    (:119-153, :266-283).  Returns (real [(idx1, idx2)], null [(idx1, idx2)],
    group sizes of half 1 / half 2).  For task variants the null split is
    composed with the full row permutation of X (:282-283)."""
    tables = resample.subject_tables(cond_order)
    alltab = np.concatenate(tables)
    nc = alltab.shape[1]
    real = []
    g1 = g2 = None
    for _ in range(num_split):
        p1, p2, g1, g2 = [], [], [], []
        for tbl in tables:
            ns = tbl.shape[0]
            half = int(np.floor(ns / 2))
            t = tbl[np.random.permutation(ns), :]
            p1.append(t[:half, :].flatten())
            p2.append(t[half:, :].flatten())
            g1.append(len(p1[-1]) // nc)
            g2.append(len(p2[-1]) // nc)
        real.append((np.concatenate(p1), np.concatenate(p2)))
    null = []
    half = sum(g1)
    for _ in range(num_split):
        t = alltab[np.random.permutation(n // nc), :]
        i1, i2 = t[:half, :].flatten(), t[half:, :].flatten()
        if task:
            perm = np.random.permutation(n)
            i1, i2 = perm[i1], perm[i2]
        null.append((i1, i2))
    return real, null, g1, g2


def _mct_items(cond_order, mctype, n, pairs, g1, g2):
    """Stacked operator rows [W1 P1; W2 P2] (2k x n) for every split."""
    nc = np.asarray(cond_order).shape[1]
    W1 = operators.mean_centre_operator(_get_cond_order((sum(g1) * nc,), tuple(g1), nc), mctype)
    W2 = operators.mean_centre_operator(_get_cond_order((sum(g2) * nc,), tuple(g2), nc), mctype)
    k = W1.shape[0]
    rows = np.zeros((len(pairs), 2 * k, n))
    for i, (i1, i2) in enumerate(pairs):
        rows[i, :k, i1] = W1.T          # half-1 row r of the gathered block is X[i1[r]]
        rows[i, k:, i2] = W2.T
    return rows, k


def _decompose(engine, rows, k):
    """Per item: U1, s1, U2, s2 (NumPy) and G12."""
    rank, nranks = dist.world()
    S = rows.shape[0]
    lo, hi = dist.shard_bounds(S, rank, nranks)
    G = engine.gram_phase(rows[lo:hi]) if hi > lo else None
    if G is None:
        import torch
        mm = (2 * k + 15) // 16 * 16
        G = torch.zeros((0, mm, mm), dtype=torch.float64, device=engine.device)
        e1 = e2 = torch.zeros((0, k), dtype=torch.float64, device=engine.device)
        v1 = v2 = torch.zeros((0, k, k), dtype=torch.float64, device=engine.device)
    else:
        e1, v1 = engine.eigh(G, 0, k)
        e2, v2 = engine.eigh(G, k, k)
    G12 = G[:, :k, k:2 * k].contiguous()
    (e1, v1, e2, v2, G12), _ = dist.exchange([e1, v1, e2, v2, G12], [], S)
    e1, v1, e2, v2, G12 = (t.cpu().numpy() for t in (e1, v1, e2, v2, G12))
    tol = 64 * k * np.finfo(float).eps

    def sv(e):
        live = e > tol * np.maximum(e[:, :1], 0.0)
        return np.sqrt(np.where(live, e, 0.0))
    return v1, sv(e1), v2, sv(e2), G12


def _inv(s):
    with np.errstate(divide="ignore"):
        return np.where(s > 0, 1.0 / s, 0.0)


def _prepare(pls_alg, matrix, cond_order, num_split, mctype, engine):
    if pls_alg != "mct":
        raise exceptions.NotImplementedError(f"split-half for {pls_alg} is not available yet")
    cond_order = np.asarray(cond_order)
    n, p = matrix.shape
    engine = engine if engine is not None else ProjectionEngine(matrix)
    real, null, g1, g2 = _draw_splits(cond_order, num_split, n, task=True)
    rows_r, k = _mct_items(cond_order, mctype, n, real, g1, g2)
    rows_0, _ = _mct_items(cond_order, mctype, n, null, g1, g2)
    if p < k:
        raise exceptions.NotImplementedError("split-half with fewer voxels than latent variables")
    return engine, np.concatenate((rows_r, rows_0)), k


def split_half_test_train(pls_alg, matrix, Y, cond_order, num_split, mctype=None, contrasts=None,
                          bscan=None, Xbscan=None, Ybscan=None, engine=None):
    """split_half_resampling.py:23-401."""
    engine, rows, k = _prepare(pls_alg, matrix, cond_order, num_split, mctype, engine)
    U1, s1, _, _, G12 = _decompose(engine, rows, k)
    train = np.repeat(s1[:, None, :], k, axis=1)                       # :195 (row broadcast, Q11)
    test = _inv(s1)[:, :, None] * np.einsum("sji,sjl,slm->sim", U1, G12, U1)    # :196
    S = num_split

    def slab(a):
        return np.transpose(a, (1, 2, 0))
    tr, te = slab(train[:S]), slab(test[:S])
    tr0, te0 = slab(train[S:]), slab(test[S:])
    with np.errstate(divide="ignore", invalid="ignore"):
        return {
            "pls_s_train": tr,
            "pls_s_test": te,
            "z": [np.mean(te[i, i, :]) / np.std(te[i, i, :], ddof=1) for i in range(k)],       # :390-393
            "pls_s_train_null": tr0,
            "pls_s_test_null": te0,
            "z_null": [np.mean(te0[i, i, :]) / np.std(te0[i, i, :], ddof=1) for i in range(k)],
        }


def split_half(pls_alg, matrix, Y, cond_order, num_split, mctype=None, contrasts=None, bscan=None,
               Xbscan=None, Ybscan=None, lv=1, CI=0.95, engine=None):
    """split_half_resampling.py:404-861."""
    engine, rows, k = _prepare(pls_alg, matrix, cond_order, num_split, mctype, engine)
    U1, s1, U2, s2, G12 = _decompose(engine, rows, k)
    u_rep = (_inv(s1)[:, :, None] * np.einsum("sji,sjl,slm->sim", U1, G12, U2)) * _inv(s2)[:, None, :]   # :682
    v_rep = np.einsum("sji,sjm->sim", U1, U2)                                                             # :683
    S = num_split

    def slab(a):
        return np.transpose(a, (1, 2, 0))
    u, v, u0, v0 = slab(u_rep[:S]), slab(v_rep[:S]), slab(u_rep[S:]), slab(v_rep[S:])
    out = {}

    def stats(prefix, arr, tag, with_std):
        """:805-853; CI (0..1) goes straight to np.percentile like the reference (Q10)."""
        diag = [np.abs(arr[i, i, :]) for i in range(lv)]
        out[f"{prefix}_mean_{tag}"] = [np.mean(x) for x in diag]
        if with_std:
            out[f"{prefix}_std_{tag}"] = [np.std(x) for x in diag]
        with np.errstate(divide="ignore", invalid="ignore"):
            out[f"{prefix}_z_{tag}"] = [np.mean(x) / np.std(x, ddof=1) for x in diag]
        out[f"{prefix}_ul_{tag}"] = [np.percentile(x, CI) for x in diag]
        out[f"{prefix}_ll_{tag}"] = [np.percentile(x, 100 - CI) for x in diag]

    stats("pls_rep", u, "u", False)
    stats("pls_rep", v, "v", False)
    stats("pls_null", u0, "u", True)
    stats("pls_null", v0, "v", True)
    out["pls_dist_u"], out["pls_dist_v"] = u, v
    out["pls_dist_null_u"], out["pls_dist_null_v"] = u0, v0
    return out
