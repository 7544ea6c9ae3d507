"""Split-half reproducibility tests on the GPU -- drop-in for
plspy/core/split_half_resampling.py (split_half_test_train :23-401,
split_half :404-861).

Per split the reference gathers both halves of X, preprocesses them, runs a
LAPACK SVD on each (k x p) block and multiplies the singular vectors.  Here a
split is one *item* of the Gram kernel: the two halves' operators are stacked,
the kernel returns

    G = [M1; M2] [M1; M2]^T      M_h = A_h Z   (never stored)

and everything the reference derives from the SVDs follows from the k x k
blocks G11, G12, G22 and the Jacobi eigen-decompositions of G11 / G22:

    U_h, s_h^2            = eig(G_hh)
    V1^T M2^T U1          = S1^-1 U1^T G12 U1                 (:196)
    V1^T V2               = S1^-1 U1^T G12 U2 S2^-1           (:682)
    U1^T U2                                                    (:683)

Z is X itself for task PLS (the half's mean-centring and row selection are a
linear operator).  For behaviour / multiblock PLS each half's correlation block
z-scores X within the half's cells, which depends on the split, so every item
gets its own gathered + z-scored copy of the rows it needs (gather_zscore, K3)
and A_h holds the z-scored behaviour columns.  The multiblock row normalisation
over all voxels (class_functions.py:503-505) is applied to G afterwards
(G_ij / (|row_i| |row_j|), |row_i|^2 = G_ii).

Random draws follow the reference's np.random call order.  Signs of singular
vectors are arbitrary in the reference (LAPACK) and here (Jacobi); the summary
statistics use absolute values of the diagonal and are sign-free.  Latent
variables that are null by construction (rank-deficient mean-centring) have
arbitrary vectors in the reference; here their singular value is deflated to 0
and their entries are 0."""
import numpy as np
import torch

from . import class_functions as cf
from . import dist, exceptions, operators, resample
from .engine import ProjectionEngine


def _get_cond_order(X_shape, groups_tuple, num_conditions):
    """split_half_resampling.py:5-21."""
    if sum(groups_tuple) * num_conditions != X_shape[0]:
        raise exceptions.InputMatrixDimensionMismatchError(
            "Derived condition ordering not compatible with input matrix"
            "X's row count. Please specify a custom cond_order field.")
    return np.array([np.array([i] * num_conditions) for i in groups_tuple])


def _draw_splits(pls_alg, cond_order, num_split, n, bscan):
    """Every real and null split, in the reference's RNG order (:119-169,
    :266-283, :316).  Each entry: dict(x1, x2 = rows of X per half; y1, y2 =
    rows of Y per half (rb: permuted in the null, :316-318); b1, b2 = bscan rows
    per half (mb); xb1, xb2 = rows of X for the bscan part (mb: through the
    null's row permutation of X, :282-283, :358)).  Returns (splits, g1, g2).
    The shuffles come from the native generator in two calls (all real splits, then all null splits:
    a Python loop around np.random.permutation cost 15 us per split); the tables are formed for all
    splits at once."""
    tables = resample.subject_tables(cond_order)
    alltab = np.concatenate(tables)
    nc = alltab.shape[1]
    mb = pls_alg in ("mb", "cmb")
    S = num_split
    halves = [int(np.floor(t.shape[0] / 2)) for t in tables]
    g1 = [h for h in halves]
    g2 = [t.shape[0] - h for t, h in zip(tables, halves)]
    perms = resample.permutation_rounds([t.shape[0] for t in tables], S)             # :136, group by group
    p1, p2, q1, q2 = [], [], [], []
    for tbl, pm, half in zip(tables, perms, halves):
        t = tbl[pm]                                                                   # (S, ns, nc)
        p1.append(t[:, :half].reshape(S, -1))
        p2.append(t[:, half:].reshape(S, -1))
        if mb:
            q1.append(t[:, :half][:, :, bscan].reshape(S, -1))                        # :160-161
            q2.append(t[:, half:][:, :, bscan].reshape(S, -1))
    x1, x2 = np.concatenate(p1, axis=1), np.concatenate(p2, axis=1)
    cols = dict(x1=x1, x2=x2, y1=x1, y2=x2)
    if mb:
        b1, b2 = np.concatenate(q1, axis=1), np.concatenate(q2, axis=1)
        cols.update(b1=b1, b2=b2, xb1=b1, xb2=b2)
    real = cols
    half = sum(g1)                                                   # :270 (the real splits' sizes, quirk Q12)
    subj, rowp = resample.permutation_rounds([n // nc, n], S)                         # :271, then :282 / :316
    t = alltab[subj]                                                                  # (S, nsub, nc)
    i1, i2 = t[:, :half].reshape(S, -1), t[:, half:].reshape(S, -1)
    null = dict(x1=i1, x2=i2, y1=i1, y2=i2)
    if mb:
        null.update(b1=t[:, :half][:, :, bscan].reshape(S, -1), b2=t[:, half:][:, :, bscan].reshape(S, -1))
    rows = np.arange(S)[:, None]
    if pls_alg in ("mct", "cst", "mb", "cmb"):
        null.update(x1=rowp[rows, i1], x2=rowp[rows, i2])
        if mb:
            null.update(xb1=rowp[rows, null["b1"]], xb2=rowp[rows, null["b2"]])      # :358: permx rows, unpermuted Y rows
    else:
        null.update(y1=rowp[rows, i1], y2=rowp[rows, i2])
    out = _SplitTable({key: np.concatenate((real[key], null[key])) for key in real})
    return out, g1, g2


class _SplitTable:
    """The drawn splits as one (2 S, rows) int array per key; indexing gives a split's dict, slicing or an
    index array a sub-table (the per-split dicts of round 2 cost a Python loop per use)."""

    def __init__(self, cols):
        self.cols = cols

    def __len__(self):
        return len(next(iter(self.cols.values())))

    def take(self, sel):
        return _SplitTable({k: v[sel] for k, v in self.cols.items()})

    def stack(self, key):
        return self.cols[key]


def _task_operator(co, mctype, centre):
    """Mean-centring operator (mct / mb) or plain cell means (cst / cmb,
    split_half_resampling.py:212, class_functions.py:482)."""
    return operators.mean_centre_operator(co, mctype) if centre else operators.cell_mean_operator(co)


def _items_mct(cond_order, mctype, n, splits, g1, g2, centre=True):
    """Stacked operator rows [W1 P1; W2 P2] (2k x n) for every split; Z = X."""
    nc = np.asarray(cond_order).shape[1]
    W1 = _task_operator(_get_cond_order((sum(g1) * nc,), tuple(g1), nc), mctype, centre)
    W2 = _task_operator(_get_cond_order((sum(g2) * nc,), tuple(g2), nc), mctype, centre)
    k = W1.shape[0]
    S = len(splits)
    rows = np.zeros((S, 2 * k, n))
    ridx = np.arange(S)[:, None]
    # half-1 row r of the gathered block is X[x1[r]]: column x1[r] of the operator gets W1[:, r]
    rows[ridx, :k, splits.stack("x1")] = W1.T[None]
    rows[ridx, k:, splits.stack("x2")] = W2.T[None]
    return rows, None, k


def _items_rb(cond_order, Y, splits, g1, g2):
    """Behaviour PLS: item matrix = [half 1 rows; half 2 rows] z-scored within
    each half's cells; operator rows = the halves' z-scored behaviour columns
    (R_h = Yz_h.T @ Xz_h, class_functions.py:185-247 via :203-204)."""
    nc = np.asarray(cond_order).shape[1]
    co1 = _get_cond_order((sum(g1) * nc,), tuple(g1), nc)
    co2 = _get_cond_order((sum(g2) * nc,), tuple(g2), nc)
    b1, b2 = cf.cell_bounds(co1), cf.cell_bounds(co2)
    n1 = int(b1[-1])
    n = n1 + int(b2[-1])
    Y = np.asarray(Y, dtype=float)
    k = (len(b1) - 1) * Y.shape[1]
    S = len(splits)
    nbeh = Y.shape[1]
    rows = np.zeros((S, 2 * k, n))
    src = np.concatenate((splits.stack("x1"), splits.stack("x2")),
                         axis=1).astype(np.int32)
    # all splits at once: z-score the halves' behaviour rows within cells and lay
    # them out as block-diagonal operator rows
    for h, (key, bb, coff, roff) in enumerate((("y1", b1, 0, 0), ("y2", b2, n1, k))):
        Yz = cf.zscore_cells(Y[splits.stack(key)], bb)          # S x n_h x b
        for c, (lo, hi) in enumerate(zip(bb[:-1], bb[1:])):
            rows[:, roff + c * nbeh:roff + (c + 1) * nbeh, coff + lo:coff + hi] = np.transpose(Yz[:, lo:hi], (0, 2, 1))
    cell_lo = np.concatenate((b1, n1 + b2[1:]))
    gather = dict(src=src, cell_lo=cell_lo, cell_z=np.ones(len(cell_lo) - 1, dtype=np.int32))
    return rows, gather, k


def _items_mb(cond_order, mctype, Y, bscan, splits, g1, g2, centre=True):
    """Multiblock PLS: item matrix = [half-1 rows, half-2 rows (raw) ; half-1
    bscan rows, half-2 bscan rows (z-scored within the half's bscan cells)];
    operator rows per half and group = [mean-centring rows ; z-scored behaviour
    columns] (class_functions.py:454-516 without the row normalisation, which is
    applied to the Gram)."""
    cond_order = np.asarray(cond_order)
    ng, nc = cond_order.shape
    nbs = len(bscan)
    Y = np.asarray(Y, dtype=float)
    b = Y.shape[1]
    per = nc + nbs * b
    k = ng * per
    halves = []
    for gs in (g1, g2):
        co = _get_cond_order((sum(gs) * nc,), tuple(gs), nc)
        halves.append(dict(W=_task_operator(co, mctype, centre), n=int(co.sum()),
                           bb=cf.cell_bounds(co[:, bscan])))
    n1, n2 = halves[0]["n"], halves[1]["n"]
    nb1, nb2 = int(halves[0]["bb"][-1]), int(halves[1]["bb"][-1])
    n = n1 + n2
    width = n + nb1 + nb2
    S = len(splits)
    rows = np.zeros((S, 2 * k, width))
    src = np.concatenate([splits.stack(key) for key in ("x1", "x2", "xb1", "xb2")],
                         axis=1).astype(np.int32)
    for h, (xoff, boff, key) in enumerate(((0, n, "b1"), (n1, n + nb1, "b2"))):
        hv = halves[h]
        Yz = cf.zscore_cells(Y[splits.stack(key)], hv["bb"])     # S x nb_h x b, all splits at once
        for g in range(ng):
            r0 = h * k + g * per
            rows[:, r0:r0 + nc, xoff:xoff + hv["n"]] = hv["W"][g * nc:(g + 1) * nc]
            for ci in range(nbs):
                lo, hi = hv["bb"][g * nbs + ci], hv["bb"][g * nbs + ci + 1]
                rb0 = r0 + nc + ci * b
                rows[:, rb0:rb0 + b, boff + lo:boff + hi] = np.transpose(Yz[:, lo:hi], (0, 2, 1))
    cell_lo = np.concatenate(([0, n], n + halves[0]["bb"][1:], n + nb1 + halves[1]["bb"][1:]))
    cell_z = np.ones(len(cell_lo) - 1, dtype=np.int32)
    cell_z[0] = 0                                    # the task rows are used raw
    return rows, dict(src=src, cell_lo=cell_lo, cell_z=cell_z), k


def _cells_rb(cond_order, splits, g1, g2):
    """Cell description of the behaviour-PLS items for the two-stage Gram kernel (engine.split_gram): per
    half, the group x condition cells of the half's row vector (consecutive blocks, split_half_resampling.py
    :172-173, :203-204 -- read with the HALF's cond_order, like the reference), every one with b behaviour
    rows.  Row l of the stacked cross-block [R1; R2] is behaviour l % b of cell l // b."""
    nc = np.asarray(cond_order).shape[1]
    xs, ys, rows = [], [], []
    for gs, xk, yk in ((g1, "x1", "y1"), (g2, "x2", "y2")):
        bb = cf.cell_bounds(_get_cond_order((sum(gs) * nc,), tuple(gs), nc))
        xs.append(splits.stack(xk))
        ys.append(splits.stack(yk))
        rows += [int(hi - lo) for lo, hi in zip(bb[:-1], bb[1:])]
    return dict(xsrc=np.concatenate(xs, axis=1), ysrc=np.concatenate(ys, axis=1), cell_rows=rows, nbq=len(rows),
                Wc=None, normalise=False)


def _cells_mb(cond_order, mctype, bscan, splits, g1, g2, centre=True):
    """Cell description of the multiblock items: behaviour cells first (per half, group and bscan condition:
    consecutive blocks of the half's bscan row vector, :160-161, :245-251), then the task cells (per half,
    group and condition: blocks of the half's row vector) with the task rows' coefficients Wc -- the
    mean-centring operator (or the plain cell means of cmb) is constant inside a cell.  None when it is not."""
    cond_order = np.asarray(cond_order)
    ng, nc = cond_order.shape
    nbs = len(bscan)
    xs, ys, rows_b, rows_t, Ws = [], [], [], [], []
    for gs, xk, bk in ((g1, "xb1", "b1"), (g2, "xb2", "b2")):
        co = _get_cond_order((sum(gs) * nc,), tuple(gs), nc)
        bb = cf.cell_bounds(co[:, bscan])
        xs.append(splits.stack(xk))
        ys.append(splits.stack(bk))
        rows_b += [int(hi - lo) for lo, hi in zip(bb[:-1], bb[1:])]
    nbq = len(rows_b)
    for gs, xk in ((g1, "x1"), (g2, "x2")):
        co = _get_cond_order((sum(gs) * nc,), tuple(gs), nc)
        cb = cf.cell_bounds(co)
        W = _task_operator(co, mctype, centre)                         # (ng nc) x n_h
        Wcell = W[:, cb[:-1]]
        if not np.array_equal(W, np.repeat(Wcell, np.diff(cb), axis=1)):
            return None
        xs.append(splits.stack(xk))
        ys.append(np.zeros_like(xs[-1]))
        rows_t += [int(hi - lo) for lo, hi in zip(cb[:-1], cb[1:])]
        Ws.append(Wcell)
    nq = nbq + len(rows_t)
    kt = ng * nc
    Wc = np.zeros((2 * kt, nq))
    Wc[:kt, nbq:nbq + ng * nc] = Ws[0]
    Wc[kt:, nbq + ng * nc:] = Ws[1]
    return dict(xsrc=np.concatenate(xs, axis=1), ysrc=np.concatenate(ys, axis=1), cell_rows=rows_b + rows_t,
                nbq=nbq, Wc=Wc, normalise=True, ng=ng, nc=nc, nbs=nbs)


def _cell_row_map(cells, k, b):
    """row_cell / row_sub of engine.split_gram for the stacked cross-block's 2k rows."""
    row_cell, row_sub = [], []
    if cells["Wc"] is None:
        for l in range(2 * k):
            row_cell.append(l // b)
            row_sub.append(l % b)
    else:
        ng, nc, nbs = cells["ng"], cells["nc"], cells["nbs"]
        per = nc + nbs * b
        for h in range(2):
            for g in range(ng):
                for r in range(per):
                    if r < nc:
                        row_cell.append(-1)
                        row_sub.append((h * ng + g) * nc + r)
                    else:
                        row_cell.append((h * ng + g) * nbs + (r - nc) // b)
                        row_sub.append((r - nc) % b)
    return row_cell, row_sub


# A singular value whose square is below this fraction of the largest one is refined (see _decompose):
# the eigen-decomposition of a Gram loses eps (s_1 / s_i)^2 of relative accuracy, 2e-11 at this ratio
REFINE_RATIO = 1e-5
DECOMPOSE_BATCHES = 1          # (2 measured +1 %, 4 -4 %, 8 -18 %: smaller launches cost what the overlap gains, microbench/split_batches.py)


def _grams(engine, item, sel):
    """(len(sel), mm, mm) Grams of the stacked cross-blocks of items `sel` on the device, the multiblock row
    normalisation (class_functions.py:503-505: G_ij / (|row_i| |row_j|)) applied; and the norms of the
    un-normalised rows ((len(sel), 2k) NumPy, or None when nothing is normalised)."""
    cells = item.get("cells")
    if cells is not None:
        sub = dict(cells, xsrc=cells["xsrc"][sel], ysrc=cells["ysrc"][sel])
        res = engine.split_gram(sub, item["Y"])
        if res is not None:
            G, rown = res
            return G, (rown[:, :2 * item["k"]] if rown is not None else None)
    rows, gather = item["dense"](sel)
    G = engine.gram_phase(rows, gather=gather)
    if not item["row_normalise"]:
        return G, None
    Gh = G.cpu().numpy()
    d = np.sqrt(np.einsum("sii->si", Gh))
    with np.errstate(divide="ignore", invalid="ignore"):
        Gh = np.where((d[:, :, None] * d[:, None, :]) > 0, Gh / d[:, :, None] / d[:, None, :], 0.0)
    return torch.as_tensor(Gh, device=engine.device), torch.as_tensor(d[:, :2 * item["k"]], device=engine.device)


def _structural_nulls(item):
    """(nn1, nn2): how many singular values of each half's cross-block vanish for ANY data -- k minus the rank
    of the half's rows as combinations of rows of X (a rank-deficient mean-centring, more behaviours than a
    cell has degrees of freedom).  From the first item's stacked operator; these latent variables come out
    of LAPACK as noise in the reference and as exact zeros here."""
    rows, gather = item["dense"](np.array([0]))
    k = item["k"]
    rows = np.asarray(rows[0], dtype=float)
    if gather is not None:                       # columns that read the same row of X act together
        src = np.asarray(gather["src"][0])
        agg = np.zeros((rows.shape[0], int(src.max()) + 1))
        np.add.at(agg.T, src, rows.T)
        rows = agg
    out = []
    for blk in (rows[:k], rows[k:2 * k]):
        out.append(k - min(int(np.linalg.matrix_rank(blk)), item["p"]))
    return tuple(out)


def _refine(engine, item, sel, v1, v2, rown, passes, second=True):
    """Refinement passes for the items `sel` (global ids) whose first-pass bases are v1 / v2 (len(sel), k, k):
    the stacked operator rows are expressed in those bases (the halves' cross-blocks then have nearly
    orthogonal rows with norms close to the singular values), their Gram is formed again -- entry (i, j) now
    carries an error of eps s_i s_j instead of eps s_1^2 -- and Jacobi continues in relative mode from the
    accumulated basis (engine.thin_svd_device does the same for the observed decomposition).  Returns
    (e1, v1, e2, v2, H) as NumPy with H = v1^T G12 v2 in the refined bases."""
    k = item["k"]
    rows, gather = item["dense"](sel)
    rows = np.array(rows, dtype=float)
    if rown is not None:                         # the multiblock row normalisation, on the operator rows
        with np.errstate(divide="ignore"):
            rows *= np.where(rown > 0, 1.0 / rown, 0.0)[:, :, None]
    for _ in range(passes):
        R = np.concatenate((np.transpose(v1, (0, 2, 1)) @ rows[:, :k], np.transpose(v2, (0, 2, 1)) @ rows[:, k:2 * k]),
                           axis=1)
        G2 = engine.gram_phase(R, gather=gather)
        e1, w1 = engine.eigh(G2, 0, k, init=engine.dev(np.ascontiguousarray(v1)), relative=True)
        if second:
            e2, w2 = engine.eigh(G2, k, k, init=engine.dev(np.ascontiguousarray(v2)), relative=True)
            e1, w1, e2, w2, H0 = engine.fetch_async([e1, w1, e2, w2, G2[:, :k, k:2 * k].contiguous()]).get()
        else:                                        # (the second half keeps its basis: nothing is asked of it)
            e1, w1, H0 = engine.fetch_async([e1, w1, G2[:, :k, k:2 * k].contiguous()]).get()
            e2, w2 = np.zeros_like(e1), v2
        J1, J2 = np.transpose(v1, (0, 2, 1)) @ w1, np.transpose(v2, (0, 2, 1)) @ w2
        H = np.transpose(J1, (0, 2, 1)) @ H0 @ J2
        v1, v2 = np.array(w1), np.array(w2)
    return np.array(e1), v1, np.array(e2), v2, H


def _decompose(engine, item, contrasts=None, second=True):
    """Per item, as NumPy.  Without contrasts: (U1, s1, U2, s2, H, P) from the Jacobi eigen-decompositions of the
    Gram blocks G11 / G22, with H = U1^T G12 U2 and P = U1^T U2 (second=False: only the first half is decomposed,
    U2 = None, s2 = 0, H = U1^T G12 and P = U1^T G12 U1 -- split_half_test_train asks nothing else of the second
    half).  Items with a graded spectrum (a
    structurally live singular
    value below sqrt(REFINE_RATIO) of the largest) get refinement passes (_refine), so that every live
    singular value keeps LAPACK's accuracy; latent variables that are null for any data (_structural_nulls)
    are returned as exact zeros -- the reference thresholds nothing here (split_half_resampling.py:194-196),
    its values for them are rounding noise.
    With contrasts C (k x q), where _run_pls_contrast (class_functions.py:126-162) gives U = C, s = row
    norms of C.T M, V = (C.T M).T:  (None, s1, None, None, C.T G12 C) with s1 = sqrt(diag(C.T G11 C))."""
    rank, nranks = dist.world()
    S, k = item["S"], item["k"]
    lo, hi = dist.shard_bounds(S, rank, nranks)
    mm = (2 * k + 15) // 16 * 16
    eps = np.finfo(float).eps
    # The shard's items go to the device in a few batches, all enqueued before the host waits for anything: the
    # host's share of batch i (H, the refinement test) then runs beside the kernels of batches i + 1 ...
    sel_all = np.arange(lo, hi)
    batches = [b for b in np.array_split(sel_all, max(1, min(DECOMPOSE_BATCHES, len(sel_all) // 256))) if len(b)]
    if contrasts is not None:
        Gs = [_grams(engine, item, sel)[0][:, :2 * k, :2 * k].contiguous() for sel in batches]
        G = torch.cat(Gs) if Gs else torch.zeros((0, 2 * k, 2 * k), dtype=torch.float64, device=engine.device)
        (Gall,), _ = dist.exchange([G], [], S)
        Gall = Gall.cpu().numpy()
        C = np.asarray(contrasts, dtype=float)
        s1 = np.sqrt(np.einsum("kq,skl,lq->sq", C, Gall[:, :k, :k], C))
        return None, s1, None, None, C.T @ Gall[:, :k, k:] @ C, None
    # H = U1^T G12 U2 and the product of the bases the callers ask for -- P = U1^T U2 (split_half, :683), or with
    # second=False P = U1^T G12 U1 (split_half_test_train, :196) -- are formed on the device right behind the Jacobi
    # kernels (batched 38 x 38 products; on the host they were 25 of a phase's 150 ms at config 4); the host keeps
    # the refinement test and redoes the products of the few items it refines
    pending = []
    for sel in batches:
        G, rown = _grams(engine, item, sel)
        e1, v1 = engine.eigh(G, 0, k)
        v1t = v1.transpose(1, 2)
        G12 = G[:, :k, k:2 * k]
        if second:
            e2, v2 = engine.eigh(G, k, k)
            want = [e1, v1, v1t @ G12 @ v2, v1t @ v2, e2, v2]
        else:
            Hd = v1t @ G12
            want = [e1, v1, Hd, Hd @ v1]
        pending.append((sel, engine.fetch_async(want + ([rown] if rown is not None else [])), rown is not None))
    out = [[] for _ in range(6)]
    if pending:
        nn1, nn2 = _structural_nulls(item)           # (host work beside the kernels)
    for sel, fetch, has_rown in pending:
        got = [np.array(a) for a in fetch.get()]
        e1, v1, H, P = got[:4]
        rest = got[-1:] if has_rown else []
        if second:
            e2, v2 = got[4:6]
        else:
            e2, v2 = np.zeros_like(e1), None
        tol = np.full((len(sel), 1), 64 * k * eps)
        # graded spectra: the smallest structurally live eigenvalue against the largest
        ratio = e1[:, max(k - nn1 - 1, 0)] / np.maximum(e1[:, 0], np.finfo(float).tiny)
        if second:
            ratio = np.minimum(ratio, e2[:, max(k - nn2 - 1, 0)] / np.maximum(e2[:, 0], np.finfo(float).tiny))
        need = np.flatnonzero(ratio < REFINE_RATIO)
        if need.size:
            deep = ratio[need] < 1e-9
            eye = np.eye(k)
            for grp, passes in ((need[~deep], 1), (need[deep], 2)):
                if grp.size:
                    w2 = v2[grp] if second else np.broadcast_to(eye, (grp.size, k, k)).copy()
                    r = _refine(engine, item, sel[grp], v1[grp], w2, rest[0][grp] if rest else None, passes, second)
                    e1[grp], v1[grp], e2[grp], H[grp] = r[0], r[1], r[2], r[4]
                    if second:
                        v2[grp] = r[3]
                        P[grp] = np.transpose(r[1], (0, 2, 1)) @ r[3]
                    else:
                        P[grp] = r[4] @ r[1]
            tol[need] = (4 * k * eps) ** 2
        for e, nn in ((e1, nn1), (e2, nn2)):
            e[e <= tol * np.maximum(e[:, :1], 0.0)] = 0.0
            if nn:
                e[:, k - nn:] = 0.0
        for acc, a in zip(out, (e1, v1, e2, v2 if second else e2[:, :0], H, P)):
            acc.append(a)
    if pending:
        e1, v1, e2, v2, H, P = (np.concatenate(acc) for acc in out)
    else:
        e1, v1, e2, v2, H, P = (np.zeros(shape) for shape in ((0, k), (0, k, k), (0, k), (0, k, k) if second else (0, 0),
                                                           (0, k, k), (0, k, k)))
    if nranks > 1:
        parts = [e1, v1, e2, H, P] + ([v2] if second else [])
        send, _ = dist.exchange([engine.dev(np.ascontiguousarray(a)) for a in parts], [], S)
        got = [t.cpu().numpy() for t in send]
        e1, v1, e2, H, P = got[:5]
        v2 = got[5] if second else None
    return v1, np.sqrt(e1), v2 if second else None, np.sqrt(e2), H, P


def _inv(s):
    with np.errstate(divide="ignore"):
        return np.where(s > 0, 1.0 / s, 0.0)


def _prepare(pls_alg, matrix, Y, cond_order, num_split, mctype, bscan, engine):
    if pls_alg not in ("mct", "rb", "mb", "cst", "csb", "cmb"):
        raise exceptions.NotImplementedError(f"split-half for {pls_alg} is not available")
    cond_order = np.asarray(cond_order)
    n, p = matrix.shape
    engine = engine if engine is not None else ProjectionEngine(matrix)
    rank, nranks = dist.world()
    drawn = [None]
    if rank == 0:       # one RNG stream (rank 0's), like every other phase
        drawn = [_draw_splits(pls_alg, cond_order, num_split, n, list(bscan) if bscan is not None else None)]
    if nranks > 1:
        import torch.distributed as td
        td.broadcast_object_list(drawn, src=0)
    splits, g1, g2 = drawn[0]
    item = dict(S=len(splits), Y=Y, p=p, row_normalise=pls_alg in ("mb", "cmb"))

    def pick(sel):
        return splits.take(np.asarray(sel))
    # dense(sel): the stacked dense operators (len(sel), 2k, n') + gather table of the items `sel`, built on demand
    if pls_alg in ("mct", "cst"):
        k = _task_operator(_get_cond_order((sum(g1) * cond_order.shape[1],), tuple(g1), cond_order.shape[1]), mctype,
                           pls_alg == "mct").shape[0]
        item.update(k=k, dense=lambda sel: _items_mct(cond_order, mctype, n, pick(sel), g1, g2, centre=pls_alg == "mct")[:2])
    elif pls_alg in ("rb", "csb"):
        b = np.asarray(Y).shape[1]
        k = cond_order.size * b
        cells = _cells_rb(cond_order, splits, g1, g2)
        item.update(k=k, cells=cells, dense=lambda sel: _items_rb(cond_order, Y, pick(sel), g1, g2)[:2])
    else:
        b = np.asarray(Y).shape[1]
        k = cond_order.shape[0] * (cond_order.shape[1] + len(bscan) * b)
        cells = _cells_mb(cond_order, mctype, list(bscan), splits, g1, g2, centre=pls_alg == "mb")
        item.update(k=k, cells=cells, dense=lambda sel: _items_mb(cond_order, mctype, Y, list(bscan), pick(sel), g1, g2,
                                                                  centre=pls_alg == "mb")[:2])
    if item.get("cells") is not None:
        item["cells"]["row_cell"], item["cells"]["row_sub"] = _cell_row_map(item["cells"], item["k"], b)
    return engine, item


def split_half_test_train(pls_alg, matrix, Y, cond_order, num_split, mctype=None, contrasts=None,
                          bscan=None, Xbscan=None, Ybscan=None, engine=None):
    """split_half_resampling.py:23-401."""
    engine, item = _prepare(pls_alg, matrix, Y, cond_order, num_split, mctype, bscan, engine)
    k = item["k"]
    d = k if contrasts is None else np.asarray(contrasts).shape[1]             # :79-86
    if matrix.shape[1] < d:
        raise exceptions.NotImplementedError("split-half with fewer voxels than latent variables")
    U1, s1, _, _, H, P = _decompose(engine, item, contrasts, second=False)
    train = np.repeat(s1[:, None, :], d, axis=1)                       # :195 (row broadcast, Q11)
    if contrasts is None:
        test = _inv(s1)[:, :, None] * P                                # V1.T M2.T U1 = S1^-1 U1.T G12 U1   (:196)
    else:
        test = H                                                       # V.T @ M2.T @ U = C.T M1 M2.T C (:220)
    S = num_split

    def slab(a):
        return np.transpose(a, (1, 2, 0))
    tr, te = slab(train[:S]), slab(test[:S])
    tr0, te0 = slab(train[S:]), slab(test[S:])
    with np.errstate(divide="ignore", invalid="ignore"):
        return {
            "pls_s_train": tr,
            "pls_s_test": te,
            "z": [np.mean(te[i, i, :]) / np.std(te[i, i, :], ddof=1) for i in range(d)],       # :390-393
            "pls_s_train_null": tr0,
            "pls_s_test_null": te0,
            "z_null": [np.mean(te0[i, i, :]) / np.std(te0[i, i, :], ddof=1) for i in range(d)],
        }


def split_half(pls_alg, matrix, Y, cond_order, num_split, mctype=None, contrasts=None, bscan=None,
               Xbscan=None, Ybscan=None, lv=1, CI=0.95, engine=None):
    """split_half_resampling.py:404-861."""
    engine, item = _prepare(pls_alg, matrix, Y, cond_order, num_split, mctype, bscan, engine)
    U1, s1, U2, s2, H, P = _decompose(engine, item, contrasts)
    if contrasts is None:
        u_rep = (_inv(s1)[:, :, None] * H) * _inv(s2)[:, None, :]       # V1.T V2 = S1^-1 U1.T G12 U2 S2^-1  (:682)
        v_rep = P                                                       # U1.T U2                             (:683)
    else:
        C = np.asarray(contrasts, dtype=float)
        u_rep = H                                                      # V1.T @ V2 = C.T M1 M2.T C  (:682)
        v_rep = np.broadcast_to(C.T @ C, (H.shape[0],) + (C.shape[1],) * 2)     # U1.T @ U2 = C.T C (:683)
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
