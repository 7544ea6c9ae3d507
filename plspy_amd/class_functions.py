"""Host-side helpers on SMALL matrices (n x k latents, n x b behaviour data,
k x k singular vectors).  Everything that touches the n x p data goes through
the GPU engine; these NumPy functions only serve the observed-decomposition
bookkeeping and the tiny per-resample tables, with the reference's conventions
(plspy/core/class_functions.py)."""
import warnings

import numpy as np


def cell_bounds(cond_order):
    """Row ranges of the group x condition cells, in row order."""
    lo = [0]
    for sizes in np.asarray(cond_order):
        for sz in sizes:
            lo.append(lo[-1] + int(sz))
    return np.array(lo, dtype=np.int64)


def group_stds(M, cond_order):
    """np.std (ddof 0) of every column within each group's rows
    (class_functions.py:314-368 with return_std=True; also used by the
    degenerate-behaviour guard, which slices with the FULL cond_order even on
    the bscan subset -- quirk Q8 -- hence the tolerant slicing)."""
    cond_order = np.asarray(cond_order)
    out = np.empty((len(cond_order), M.shape[-1]))
    start = 0
    with warnings.catch_warnings(), np.errstate(all="ignore"):
        warnings.simplefilter("ignore")
        for g, tot in enumerate(cond_order.sum(axis=1)):
            out[g] = np.std(M[start:start + tot], axis=0)
            start += int(tot)
    return out


def any_group_std_zero(Mb, cond_order):
    """Batched form of ``(group_stds(M, cond_order) == 0).any()`` for a stack
    Mb (m, rows, b): True where some column is constant within some group's
    rows (the degenerate-behaviour guard, bootstrap_permutation.py:349, :563).
    Same tolerant slicing as group_stds: groups beyond the rows of M are empty
    (std = nan, never == 0)."""
    cond_order = np.asarray(cond_order)
    bad = np.zeros(Mb.shape[0], dtype=bool)
    start = 0
    for tot in cond_order.sum(axis=1):
        blk = Mb[:, start:start + int(tot)]
        start += int(tot)
        if blk.shape[1] == 0:
            continue
        # np.std == 0 exactly when every deviation from the mean is exactly 0, which takes a constant
        # column (but a constant column's mean may round away from the constant: the std decides).
        # Constant columns are rare, so the two-pass std runs on the candidates only.
        cand = (blk == blk[:, :1]).all(axis=1).any(axis=-1)
        if cand.any():
            at = np.flatnonzero(cand)
            bad[at] |= (np.std(blk[at], axis=1) == 0).any(axis=-1)
    return bad


def degenerate_guard(Ysrc, cond_order):
    """is_bad(rows) for a stack of index vectors rows (m, n): any_group_std_zero(Ysrc[rows], cond_order)
    without gathering m x n x b numbers when the answer can be read off the indices.  If no column of
    Ysrc holds the same value twice, a group's column is constant only when the group repeats ONE source
    row throughout (never for a permutation, rarely for a bootstrap): those candidates are found on the
    integer table, and only they go through the exact test.  Columns with repeated values (or NaNs:
    np.unique folds them) take the full test."""
    Ysrc = np.asarray(Ysrc)
    cond_order = np.asarray(cond_order)
    distinct = all(len(np.unique(Ysrc[:, c])) == Ysrc.shape[0] for c in range(Ysrc.shape[1]))

    def is_bad(rows):
        if not distinct:
            return any_group_std_zero(Ysrc[rows], cond_order)
        cand = np.zeros(rows.shape[0], dtype=bool)
        start = 0
        for tot in cond_order.sum(axis=1):
            blk = rows[:, start:start + int(tot)]
            start += int(tot)
            if blk.shape[1]:
                cand |= (blk == blk[:, :1]).all(axis=1)
        bad = np.zeros(rows.shape[0], dtype=bool)
        if cand.any():
            at = np.flatnonzero(cand)
            bad[at] = any_group_std_zero(Ysrc[rows[at]], cond_order)
        return bad
    return is_bad


def zscore_cells(M, bounds):
    """Per-cell z-score (ddof 0) divided by sqrt(n_cell), constant columns -> 0:
    what class_functions.py:221-238 does to X and to Y (scipy.stats.zscore's
    constant-slice rule + nan_to_num).  M: (..., n, b); cells along axis -2."""
    out = np.zeros_like(M, dtype=float)
    eps = np.finfo(float).eps
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        blk = M[..., lo:hi, :]
        mu = blk.mean(axis=-2, keepdims=True)
        sd = np.sqrt(np.mean((blk - mu) ** 2, axis=-2, keepdims=True))
        with np.errstate(invalid="ignore", divide="ignore"):
            z = (blk - mu) / sd / np.sqrt(hi - lo)
        dead = np.broadcast_to(~(sd > eps * np.abs(mu)), z.shape)
        out[..., lo:hi, :] = np.where(dead, 0.0, z)
    return out


def lvcorr_from_latents(Lt, Yz, bounds):
    """Bootstrap LVcorr for a batch: Lt (cnt, k, n) = the resampled rows of
    (X @ V_hat)^T, Yz (cnt, n, b) per-cell z-scored behaviour -> (cnt, cells*b, k),
    i.e. corr_rows(Lt^T, Yz) without the transposes.  The per-cell z-score is
    scale invariant, so the column normalisation of V_hat (:623) may be skipped:
    a zero column gives a constant (zero) latent, which z-scores to 0 either way."""
    out = []
    eps = np.finfo(float).eps
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        blk = Lt[:, :, lo:hi]
        mu = blk.mean(axis=-1, keepdims=True)
        d = blk - mu
        sd = np.sqrt(np.mean(d * d, axis=-1, keepdims=True))
        with np.errstate(invalid="ignore", divide="ignore"):
            z = d / (sd * np.sqrt(hi - lo))
        z[np.broadcast_to(~(sd > eps * np.abs(mu)), z.shape)] = 0.0
        out.append(np.swapaxes(z @ Yz[:, lo:hi], 1, 2))          # (cnt, k, nc)(cnt, nc, b) -> cnt, b, k
    return np.concatenate(out, axis=1)


def compute_corr_small(L, Y, cond_order):
    """_compute_corr (class_functions.py:185-247) for a small left matrix
    (latent scores n x k): stacked per-cell Yz.T @ Lz, (cells*b) x k."""
    bounds = cell_bounds(cond_order)
    Lz = zscore_cells(np.asarray(L, dtype=float), bounds)
    Yz = zscore_cells(np.asarray(Y, dtype=float), bounds)
    return np.vstack([Yz[lo:hi].T @ Lz[lo:hi] for lo, hi in zip(bounds[:-1], bounds[1:])])


def corr_operator(Yz, bounds):
    """(cells*b) x n operator A with  A @ Xz == stacked per-cell Yz.T @ Xz,
    Xz being the per-cell z-scored data (the behaviour half of _compute_corr,
    class_functions.py:240-242, as a block-diagonal matrix)."""
    n, b = Yz.shape
    ncell = len(bounds) - 1
    A = np.zeros((ncell * b, n))
    for c, (lo, hi) in enumerate(zip(bounds[:-1], bounds[1:])):
        A[c * b:(c + 1) * b, lo:hi] = Yz[lo:hi].T
    return A


def normalize(M):
    """Column L2 normalisation, zero columns stay zero (class_functions.py:693-708)."""
    base = np.linalg.norm(M, axis=0)
    if np.any(base == 0):
        warnings.warn("_normalize: encountered column(s) with zero norm; "
                      "these will be returned as zero vectors.", RuntimeWarning)
    out = np.zeros_like(M, dtype=float)
    np.divide(M, base, out=out, where=base != 0)
    return out


def compute_Y_latents(Y, U, cond_order):
    """class_functions.py:250-276: per cell, Y_cell @ U_cell."""
    bounds = cell_bounds(cond_order)
    b = Y.shape[1]
    out = np.empty((Y.shape[0], U.shape[1]))
    for c, (lo, hi) in enumerate(zip(bounds[:-1], bounds[1:])):
        out[lo:hi] = Y[lo:hi] @ U[c * b:(c + 1) * b]
    return out


def bscan_mask(cond_order, bscan):
    """Rows of X / Y that belong to the conditions in bscan (pls_classes.py:1430-1441)."""
    mask = []
    for sizes in np.asarray(cond_order):
        for ci, sz in enumerate(sizes):
            mask.extend([ci in bscan] * int(sz))
    return np.array(mask, dtype=bool)


def smeanmat_rows(L, cond_order, mctype):
    """resample._calculate_smeanmat (resample.py:224-287) on a small matrix.  It
    is a row operator (subtracts group / condition / grand means of the rows),
    so smeanmat(X_new) @ V == smeanmat(X_new @ V): the engine applies it to the
    n x k latent scores instead of the n x p data.  L: (..., n, k)."""
    cond_order = np.asarray(cond_order)
    ng, nc = cond_order.shape
    bounds = cell_bounds(cond_order)
    gb = np.concatenate(([0], np.cumsum(cond_order.sum(axis=1))))
    cellmean = np.stack([L[..., lo:hi, :].mean(axis=-2) for lo, hi in zip(bounds[:-1], bounds[1:])], axis=-2)
    out = np.array(L, dtype=float, copy=True)
    if mctype == 0:
        for g in range(ng):
            out[..., gb[g]:gb[g + 1], :] -= L[..., gb[g]:gb[g + 1], :].mean(axis=-2, keepdims=True)
    elif mctype == 1:
        cm = np.stack([cellmean[..., [c + g * nc for g in range(ng)], :].mean(axis=-2) for c in range(nc)], axis=-2)
        for g in range(ng):
            for c in range(nc):
                lo, hi = bounds[g * nc + c], bounds[g * nc + c + 1]
                out[..., lo:hi, :] -= cm[..., c:c + 1, :]
    elif mctype == 2:
        out -= L.mean(axis=-2, keepdims=True)
    elif mctype == 3:
        cm = np.stack([cellmean[..., [c + g * nc for g in range(ng)], :].mean(axis=-2) for c in range(nc)], axis=-2)
        grand = cm.mean(axis=-2, keepdims=True)
        for g in range(ng):
            gm = L[..., gb[g]:gb[g + 1], :].mean(axis=-2, keepdims=True)
            for c in range(nc):
                lo, hi = bounds[g * nc + c], bounds[g * nc + c + 1]
                out[..., lo:hi, :] += grand - gm - cm[..., c:c + 1, :]
    else:
        raise ValueError("mctype must be 0..3")
    return out


def cell_means_rows(L, cond_order):
    """group x condition means of the rows of a small matrix (..., n, k)."""
    bounds = cell_bounds(cond_order)
    return np.stack([L[..., lo:hi, :].mean(axis=-2) for lo, hi in zip(bounds[:-1], bounds[1:])], axis=-2)


def corr_rows(L, Yz, bounds):
    """Stacked per-cell Yz.T @ zscore(L) for batches: L (R, n, k), Yz (R, n, b)
    already z-scored -> (R, cells*b, k)  (class_functions.py:185-247 on latents)."""
    Lz = zscore_cells(L, bounds)
    return np.concatenate([np.swapaxes(Yz[:, lo:hi], 1, 2) @ Lz[:, lo:hi]
                           for lo, hi in zip(bounds[:-1], bounds[1:])], axis=1)


def split_Tu_Bu(U, n_cond, n_behav, n_groups, n_bscan):
    """class_functions.py:518-578: task / behaviour rows of the multiblock U."""
    per = n_cond + n_bscan * n_behav
    Tu = np.vstack([U[g * per:g * per + n_cond] for g in range(n_groups)])
    Bu = np.vstack([U[g * per + n_cond:(g + 1) * per] for g in range(n_groups)])
    return Tu, Bu


def get_Tusc(Tu, n_cond, cond_order):
    """class_functions.py:580-625: each condition's row of Tu repeated per subject."""
    rows = []
    for g, sizes in enumerate(np.asarray(cond_order)):
        for c in range(n_cond):
            rows.append(np.tile(Tu[g * n_cond + c:g * n_cond + c + 1], (int(sizes[c]), 1)))
    return np.vstack(rows)


def get_Busc(Bu, Ybscan, cond_order, bscan):
    """class_functions.py:628-690: behaviour scores Ybscan_cell @ Bu_cell (uses the
    first condition's subject count for every condition of a group, like the
    reference)."""
    cond_order = np.asarray(cond_order)
    nb = len(bscan)
    b = Ybscan.shape[1]
    out = []
    for g, sizes in enumerate(cond_order):
        ns = int(sizes[0])
        span = int(sum(cond_order[:g, 0])) * nb
        for c in range(nb):
            out.append(Ybscan[span + ns * c:span + ns * (c + 1)] @ Bu[b * (c + nb * g):b * (c + 1 + nb * g)])
    return np.vstack(out)
