"""Small dense operators of the PLS preprocessors (host side, NumPy, k x n).

Every preprocess step the reference applies to a resampled X is linear in X
(SURVEY.md appendix A1/A2):  _mean_centre(X[inds], cond_order, mctype) ==
(W @ P) @ X  with W the k x n operator built here and P the row selection.
The engine never gathers X; it folds W, P and the observed singular vectors
into one n x k operator per resample and contracts that with X on the GPU."""
import numpy as np

from . import exceptions


def cell_slices(cond_order):
    """(start, stop) of every group x condition cell, in row order
    (reference row order: group -> condition -> subject)."""
    out = []
    start = 0
    for sizes in np.asarray(cond_order):
        for sz in sizes:
            out.append((start, start + int(sz)))
            start += int(sz)
    return out


def cell_mean_operator(cond_order):
    """Wm (g*c x n):  Wm @ X == group-condition means of X
    (class_functions.py:371-408)."""
    cells = cell_slices(cond_order)
    n = cells[-1][1]
    W = np.zeros((len(cells), n))
    for r, (a, b) in enumerate(cells):
        W[r, a:b] = 1.0 / (b - a)
    return W


def group_mean_operator(cond_order):
    """(g x n): mean over all rows of a group (class_functions.py:314-368)."""
    cond_order = np.asarray(cond_order)
    n = int(cond_order.sum())
    G = np.zeros((cond_order.shape[0], n))
    start = 0
    for g, tot in enumerate(cond_order.sum(axis=1)):
        G[g, start:start + tot] = 1.0 / tot
        start += int(tot)
    return G


def grand_condition_operator(cond_order):
    """(c x n): mean over groups of the cell means (class_functions.py:411-451)."""
    cond_order = np.asarray(cond_order)
    ng, nc = cond_order.shape
    Wm = cell_mean_operator(cond_order)
    return np.stack([Wm[[c + g * nc for g in range(ng)]].mean(axis=0) for c in range(nc)])


def mean_centre_operator(cond_order, mctype=0):
    """W (g*c x n) with  W @ X == X_mc  of class_functions.py:7-95."""
    cond_order = np.asarray(cond_order)
    ng, nc = cond_order.shape
    n = int(cond_order.sum())
    Wm = cell_mean_operator(cond_order)
    if mctype == 0:
        return Wm - np.repeat(group_mean_operator(cond_order), nc, axis=0)
    if mctype == 1:
        return Wm - np.tile(grand_condition_operator(cond_order), (ng, 1))
    if mctype == 2:
        return Wm - np.full((1, n), 1.0 / n)
    if mctype == 3:
        C = grand_condition_operator(cond_order)
        return (Wm - np.tile(C, (ng, 1)) - np.repeat(group_mean_operator(cond_order), nc, axis=0)
                + C.mean(axis=0, keepdims=True))
    raise exceptions.NotImplementedError(
        "Specified mean-centring method is either not implemented or is invalid.")


def operator_from_callable(preprocess, n, cond_order, mctype):
    """W for a caller-supplied linear preprocess with the reference's signature
    ``preprocess(X, cond_order, mctype, return_means=False)``: apply it to the
    identity."""
    return np.asarray(preprocess(np.eye(n), cond_order, mctype, return_means=False), dtype=float)
