"""CPU oracle for the plspy resampling hot path  --  TEST INFRASTRUCTURE ONLY.

This module is a plain NumPy restatement of the reference's permutation /
bootstrap / split-half algorithms (McIntosh-Lab/plspy, ``plspy/core``).  It is
the checker that the HIP path is compared against.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it; nothing under ``plspy_amd/`` does, and the product path fails loudly when
the HIP extension is missing rather than falling back to this file.

Pinning: every function here is checked against golden vectors produced by
importing the reference's own ``plspy.core`` in the build container
(``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``;
``tests/test_oracle_golden.py``).  The reference's own tests pin no values on
this path, so those generated vectors are the pin.

The oracle executes the reference's *direct* form on purpose (row gather ->
cell means / correlations -> projection, every bootstrap projection
materialised, ``np.std`` over the stack), not the operator-folded streaming
form the HIP engine uses, so that agreement between the two is evidence and
not tautology.  Random draws go through a ``Sampler`` whose default
implementation issues the same ``np.random`` legacy calls in the same order as
the reference, so seeding ``np.random`` identically reproduces the reference's
index vectors.

Citations are ``file:line`` into ``/root/reference/plspy/core``.
"""
from __future__ import annotations

import numpy as np
from scipy.stats import norm as _norm

TASK_ALGS = ("mct", "cst")
BEHAV_ALGS = ("rb", "csb")
MULTI_ALGS = ("mb", "cmb")


# --------------------------------------------------------------------------
# index draws  (resample.py, split_half_resampling.py)
# --------------------------------------------------------------------------
def subject_table(cond_order):
    """(subjects x conditions) table of row ids, groups stacked.

    Follows resample.py:44-61 / split_half_resampling.py:103-115: rows of X are
    ordered group -> condition -> subject; column ``c`` of the table holds the
    row ids of condition ``c``.
    """
    cond_order = np.asarray(cond_order)
    blocks = []
    start = 0
    for sizes in cond_order:
        cols = []
        for sz in sizes:
            cols.append(np.arange(start, start + sz))
            start += sz
        blocks.append(np.column_stack(cols))
    return blocks


class Sampler:
    """Issues the reference's ``np.random`` calls in the reference's order."""

    def perm_task(self, cond_order):
        """resample.py:63-73 ("mct","cst","mb","cmb" branch)."""
        grp = np.concatenate(subject_table(cond_order))
        # :66  one np.random.permutation per subject row (apply_along_axis)
        within = np.empty_like(grp)
        for r in range(grp.shape[0]):
            within[r] = np.random.permutation(grp[r])
        # :69-71  one permutation per condition slot over ALL subjects
        shuff = within.T.copy()
        for c in range(grp.shape[1]):
            shuff[c, :] = np.random.permutation(within.T[c, :])
        # :73  slot-major flattening (quirk Q6) -- used verbatim
        return shuff.ravel()

    def perm_rows(self, n):
        """resample.py:77 / split_half_resampling.py:136,271,282,316."""
        return np.random.permutation(n)

    def boot(self, cond_order):
        """resample.py:132-160: per group, subjects with replacement; the same
        draw is applied to every condition; flattened condition-major."""
        out = []
        for tbl in subject_table(cond_order):
            ns = tbl.shape[0]
            pick = np.random.choice(ns, ns, replace=True)   # :141
            out.append(tbl[pick, :].T.ravel())               # :143-151
        return np.concatenate(out)


class ReplaySampler(Sampler):
    """Replays a recorded list of draws (for feeding two implementations the
    same index vectors)."""

    def __init__(self, draws):
        self._draws = list(draws)
        self._pos = 0

    def _next(self):
        d = self._draws[self._pos]
        self._pos += 1
        return np.asarray(d).copy()

    def perm_task(self, cond_order):
        return self._next()

    def perm_rows(self, n):
        return self._next()

    def boot(self, cond_order):
        return self._next()


class RecordingSampler(Sampler):
    """Default draws, recorded."""

    def __init__(self):
        self.draws = []

    def perm_task(self, cond_order):
        d = super().perm_task(cond_order)
        self.draws.append(d.copy())
        return d

    def perm_rows(self, n):
        d = super().perm_rows(n)
        self.draws.append(d.copy())
        return d

    def boot(self, cond_order):
        d = super().boot(cond_order)
        self.draws.append(d.copy())
        return d


# --------------------------------------------------------------------------
# preprocess primitives  (class_functions.py)
# --------------------------------------------------------------------------
def group_condition_means(X, cond_order):
    """class_functions.py:371-408 (+ :279-311)."""
    cond_order = np.asarray(cond_order)
    out = np.empty((cond_order.size, X.shape[-1]))
    row = 0
    start = 0
    for sizes in cond_order:
        for sz in sizes:
            out[row] = np.mean(X[start:start + sz], axis=0)
            start += sz
            row += 1
    return out


def group_means(X, cond_order):
    """class_functions.py:314-368 (return_std=False)."""
    cond_order = np.asarray(cond_order)
    out = np.empty((len(cond_order), X.shape[-1]))
    start = 0
    for g, tot in enumerate(cond_order.sum(axis=1)):
        out[g] = np.mean(X[start:start + tot], axis=0)
        start += tot
    return out


def group_stds(X, cond_order):
    """class_functions.py:314-368 (return_std=True): np.std, ddof=0."""
    cond_order = np.asarray(cond_order)
    out = np.empty((len(cond_order), X.shape[-1]))
    start = 0
    for g, tot in enumerate(cond_order.sum(axis=1)):
        out[g] = np.std(X[start:start + tot], axis=0)
        start += tot
    return out


def grand_condition_means(X, cond_order):
    """class_functions.py:411-451: mean over groups of the cell means."""
    cond_order = np.asarray(cond_order)
    ng, nc = cond_order.shape
    gcm = group_condition_means(X, cond_order)
    out = np.empty((nc, X.shape[-1]))
    for c in range(nc):
        out[c] = np.mean(gcm[[c + g * nc for g in range(ng)], :], axis=0)
    return out


def mean_centre(X, cond_order, mctype=0):
    """class_functions.py:7-95.  Returns (X_means, X_mc), both (g*c, p)."""
    cond_order = np.asarray(cond_order)
    ng = cond_order.shape[0]
    X_means = group_condition_means(X, cond_order)
    if mctype == 0:       # :46-53
        gm = group_means(X, cond_order)
        reps = np.array([len(r) for r in cond_order])
        X_mc = X_means - np.repeat(gm, reps, axis=0)
    elif mctype == 1:     # :56-63
        X_mc = X_means - np.tile(grand_condition_means(X, cond_order), (ng, 1))
    elif mctype == 2:     # :66-69
        X_mc = X_means - np.mean(X, axis=0)
    elif mctype == 3:     # :73-85  (group means repeated len(cond_order[0]) times)
        gm = group_means(X, cond_order)
        cm = grand_condition_means(X, cond_order)
        grand = np.mean(cm, axis=0)
        X_mc = (X_means - np.tile(cm, (ng, 1))
                - np.repeat(gm, len(cond_order[0]), axis=0)
                + np.tile(grand, (X_means.shape[0], 1)))
    else:
        raise ValueError("mctype must be 0..3")
    return X_means, X_mc


def _zscore_cell(a):
    """scipy.stats.zscore(a) (axis 0, ddof 0) as used at class_functions.py:
    221-233, including scipy's rule that a slice whose std is <= eps*|mean|
    becomes NaN (scipy 1.15 ``zmap``), followed by the reference's
    ``nan_to_num`` (:237-238)."""
    mn = a.mean(axis=0, keepdims=True)
    sd = np.sqrt(np.mean((a - mn) ** 2, axis=0, keepdims=True))
    with np.errstate(invalid="ignore", divide="ignore"):
        z = (a - mn) / sd
    dead = sd <= np.abs(np.finfo(z.dtype).eps * mn)
    z[np.broadcast_to(dead, z.shape)] = np.nan
    return z


def compute_corr(X, Y, cond_order):
    """class_functions.py:185-247: per cell, z-score X and Y (ddof 0), divide
    each by sqrt(n_cell), NaN->0, R_cell = Yz.T @ Xz; stacked (g*c*b, p)."""
    order = np.asarray(cond_order).reshape(-1)
    nb = Y.shape[1]
    R = np.empty((order.size * nb, X.shape[1]))
    start = 0
    for i, sz in enumerate(order):
        xz = _zscore_cell(X[start:start + sz]) / np.sqrt(sz)
        yz = _zscore_cell(Y[start:start + sz]) / np.sqrt(sz)
        np.nan_to_num(xz, copy=False)
        np.nan_to_num(yz, copy=False)
        R[i * nb:(i + 1) * nb] = yz.T @ xz
        start += sz
    return R


def normalize(v):
    """class_functions.py:693-708: column L2 normalisation, zero columns stay 0."""
    base = np.linalg.norm(v, axis=0)
    out = np.zeros_like(v, dtype=float)
    np.divide(v, base, out=out, where=base != 0)
    return out


def create_multiblock(X, cond_order, pls_alg, bscan, mctype=0, norm_opt=True,
                      Xbscan=None, Ybscan=None):
    """class_functions.py:454-516."""
    cond_order = np.asarray(cond_order)
    if pls_alg == "cmb":
        task = group_condition_means(X, cond_order)          # :482
    else:
        task = mean_centre(X, cond_order, mctype)[1]          # :485
    R = compute_corr(Xbscan, Ybscan, cond_order[:, bscan])    # :488-489
    nb = Ybscan.shape[1]
    nc = cond_order.shape[1]
    nbs = len(bscan)
    out = []
    for g in range(cond_order.shape[0]):
        t = task[g * nc:(g + 1) * nc]
        r = R[g * nbs * nb:(g + 1) * nbs * nb]
        if norm_opt:                                          # :503-505
            t = t / np.linalg.norm(t, axis=1, keepdims=True)
            r = r / np.linalg.norm(r, axis=1, keepdims=True)
        out.append(np.vstack((t, r)))                         # :508
    return np.vstack(out)


def run_pls(M):
    """class_functions.py:98-123: thin SVD, returns (U, s, V) with V = Vt.T."""
    U, s, Vt = np.linalg.svd(M, full_matrices=False)
    return U, s, Vt.T


def run_pls_contrast(M, C):
    """class_functions.py:126-162: U = C, s = row norms of C.T @ M, V = (C.T @ M).T."""
    CB = C.T @ M
    return C, np.sqrt(np.sum(CB ** 2, axis=1)), CB.T


def cmb_contrast_rows(contrasts, ng, nc, nb, bscan):
    """pls_classes.py:1788-1799: rows of the user's contrast matrix kept for the
    multiblock rows (all task conditions, behaviour rows of the bscan conditions)."""
    Ti = np.ones(nc)
    Bi = np.zeros((nb, nc))
    Bi[:, bscan] = 1
    mask = np.tile(np.concatenate([Ti.reshape(-1, order="F"), Bi.reshape(-1, order="F")]), ng)
    return contrasts[mask.astype(bool), :]


def calculate_smeanmat(Xt, cond_order, mctype):
    """resample.py:224-287: row-level centring of the task bootstrap sample."""
    cond_order = np.asarray(cond_order)
    ng = cond_order.shape[0]
    if mctype == 0:
        return Xt - np.repeat(group_means(Xt, cond_order), cond_order.sum(axis=1), axis=0)
    if mctype == 1:
        cm = np.tile(grand_condition_means(Xt, cond_order), (ng, 1))
        return Xt - np.repeat(cm, cond_order.flatten(), axis=0)
    if mctype == 2:
        return Xt - np.mean(Xt, axis=0)
    if mctype == 3:
        gm = np.repeat(group_means(Xt, cond_order), cond_order.sum(axis=1), axis=0)
        cmr = grand_condition_means(Xt, cond_order)
        cm = np.repeat(np.tile(cmr, (ng, 1)), cond_order.flatten(), axis=0)
        grand = np.mean(cmr, axis=0)
        return Xt - gm - cm + np.tile(grand, (Xt.shape[0], 1))
    raise ValueError("mctype must be 0..3")


def bscan_mask(cond_order, bscan):
    """pls_classes.py:1430-1441: rows of X/Y belonging to the bscan conditions."""
    mask = []
    for sizes in np.asarray(cond_order):
        for ci, sz in enumerate(sizes):
            mask.extend([ci in bscan] * int(sz))
    return np.array(mask, dtype=bool)


# --------------------------------------------------------------------------
# observed decomposition  (pls_classes.py constructors, numbers only)
# --------------------------------------------------------------------------
def observed(pls_alg, X, cond_order, Y=None, mctype=0, bscan=None, contrasts=None):
    """The decomposition each PLS class performs before resampling.

    mct: pls_classes.py:258-266; rb: :570-586; mb: :1430-1489 (subset);
    cst: :853-865; csb: :1128-1143; cmb: :1788-1857 (subset).  For the contrast
    variants ``contrasts`` is the user's matrix (normalised here like the
    classes do) and U is the normalised contrast matrix."""
    cond_order = np.asarray(cond_order)
    out = {}
    if pls_alg in ("cst", "csb", "cmb"):
        ng, nc = cond_order.shape
        if pls_alg == "cst":
            C = normalize(contrasts)
            out["R"] = M = group_condition_means(X, cond_order)
        elif pls_alg == "csb":
            C = normalize(contrasts)
            out["R"] = M = compute_corr(X, Y, cond_order)
        else:
            C = normalize(cmb_contrast_rows(contrasts, ng, nc, Y.shape[1], bscan))
            m = bscan_mask(cond_order, bscan)
            out["Xbscan"], out["Ybscan"] = X[m], Y[m]
            out["multiblock"] = M = create_multiblock(X, cond_order, "cmb", bscan, mctype,
                                                      Xbscan=X[m], Ybscan=Y[m])
        U, s, V = run_pls_contrast(M, C)
        out["contrasts"] = C
        out["lvintercorrs"] = V.T @ V
        if pls_alg == "cst":
            out["X_latent"] = X @ normalize(V)
            out["Tvsc_orig"] = group_condition_means(out["X_latent"], cond_order)
        elif pls_alg == "csb":
            out["X_latent"] = X @ V
        else:
            Tx = X @ normalize(V)
            Bx = out["Xbscan"] @ V
            out["Tusc"], out["Busc"] = Tx, Bx
            out["Tvsc_orig"] = group_condition_means(Tx, cond_order)
            out["lvcorrs"] = compute_corr(Bx, out["Ybscan"], cond_order[:, bscan])
        out["U"], out["s"], out["V"] = U, s, V
        return out
    if pls_alg == "mct":
        out["X_means"], out["X_mc"] = mean_centre(X, cond_order, mctype)
        U, s, V = run_pls(out["X_mc"])
        out["X_latent"] = X @ V
        out["Tvsc_orig"] = group_condition_means(out["X_latent"], cond_order)
    elif pls_alg == "rb":
        out["R"] = compute_corr(X, Y, cond_order)
        U, s, V = run_pls(out["R"])
        out["X_latent"] = X @ V
        out["lvcorrs"] = compute_corr(out["X_latent"], Y, cond_order)
    elif pls_alg == "mb":
        m = bscan_mask(cond_order, bscan)
        Xb, Yb = X[m], Y[m]
        out["Xbscan"], out["Ybscan"] = Xb, Yb
        out["multiblock"] = create_multiblock(X, cond_order, "mb", bscan, mctype,
                                              Xbscan=Xb, Ybscan=Yb)
        U, s, V = run_pls(out["multiblock"])
        Tx = X @ normalize(V)
        Bx = Xb @ V
        out["Tusc"], out["Busc"] = Tx, Bx
        out["Tvsc_orig"] = group_condition_means(Tx, cond_order)
        out["lvcorrs"] = compute_corr(Bx, Yb, cond_order[:, bscan])
    else:
        raise ValueError(pls_alg)
    out["U"], out["s"], out["V"] = U, s, V
    return out


# --------------------------------------------------------------------------
# permutation test  (bootstrap_permutation.py:266-464)
# --------------------------------------------------------------------------
def permutation_test(pls_alg, X, Y, U, s, V, cond_order, mctype, niter,
                     bscan=None, Xbscan=None, Ybscan=None, threshold=1e-12,
                     sampler=None, contrast=None):
    """Returns dict(permute_ratio, stepdown_ratio, s_list, s (thresholded copy)).

    Unlike the reference (Q1) ``s`` is not mutated in place; the thresholded
    copy is returned.  ``contrast``: the (normalised) contrast matrix of the
    cst / csb / cmb variants (:407-410, :429-433)."""
    sampler = sampler or Sampler()
    cond_order = np.asarray(cond_order)
    s = np.array(s, dtype=float)
    s[np.abs(s) < threshold] = 0                               # :295
    greater = np.zeros(s.shape)
    step_greater = np.zeros(s.shape)
    s_list = np.empty((niter, s.shape[0]))
    multi = pls_alg in MULTI_ALGS

    if multi:                                                  # :305-312
        raw = create_multiblock(X, cond_order, pls_alg, bscan, mctype, norm_opt=False,
                                Xbscan=Xbscan, Ybscan=Ybscan)
        total = np.sum(raw ** 2)
        org_s = np.sqrt(s ** 2 / np.sum(s ** 2) * total)
    else:
        org_s = s.copy()                                       # :314
    tot_org = np.array([np.sum(org_s[r:] ** 2) for r in range(len(org_s))])   # :317-319

    for i in range(niter):
        if pls_alg == "mct":
            inds = sampler.perm_task(cond_order)               # :329
            permuted = mean_centre(X[inds], cond_order, mctype)[1]   # :385
        elif pls_alg == "cst":
            inds = sampler.perm_task(cond_order)               # :329
            permuted = group_condition_means(X[inds], cond_order)    # :389
        elif pls_alg in BEHAV_ALGS:
            for _ in range(100):                               # :334-353
                Yn = Y[sampler.perm_rows(Y.shape[0])]          # :338
                if not (group_stds(Yn, cond_order) == 0).any():
                    break
            else:
                raise Exception("degenerate behaviour data")   # :355
            permuted = compute_corr(X, Yn, cond_order)         # :396
        elif multi:
            for _ in range(100):
                ti = sampler.perm_task(cond_order)             # :343
                Yn = Ybscan[sampler.perm_rows(Ybscan.shape[0])]   # :347
                # :349 guard uses the FULL cond_order on the bscan subset (Q8)
                if not (group_stds(Yn, cond_order) == 0).any():
                    break
            else:
                raise Exception("degenerate behaviour data")
            Xt = X[ti]
            permuted = create_multiblock(Xt, cond_order, pls_alg, bscan, mctype,
                                         Xbscan=Xbscan, Ybscan=Yn)     # :392
        else:
            raise ValueError(pls_alg)

        if contrast is not None:                               # :429-433 (the SVD at :410 is discarded, Q4)
            cross = normalize(contrast).T @ permuted
            s_hat = np.sqrt(np.sum(cross ** 2, axis=1))
            greater += s_hat >= s
            s_list[i] = s_hat
            tot_perm = np.array([np.sum(s_hat[r:] ** 2) for r in range(len(s_hat))])
            step_greater += tot_perm >= tot_org
            continue

        VS = permuted.T @ U                                    # :404
        s_hat = np.sqrt(np.sum(VS ** 2, axis=0))               # :405

        if pls_alg == "mb":                                    # :413-427
            raw = create_multiblock(Xt, cond_order, "mb", bscan, mctype, norm_opt=False,
                                    Xbscan=Xbscan, Ybscan=Yn)
            tot_hat = np.sum(raw ** 2)
            per_hat = s_hat ** 4 / np.sum(s_hat ** 4)          # quirk Q3
            s_hat = np.sqrt(per_hat * tot_hat)
            greater += s_hat >= org_s
        else:                                                  # :435-437
            s_hat[np.abs(s_hat) < threshold] = 0
            greater += s_hat >= s
        s_list[i] = s_hat
        tot_perm = np.array([np.sum(s_hat[r:] ** 2) for r in range(len(s_hat))])
        step_greater += tot_perm >= tot_org                    # :447-451

    return {
        "permute_ratio": greater / (niter + 1),                # :444 (Q2)
        "stepdown_ratio": step_greater / (niter + 1),          # :452
        "s_list": s_list,
        "s": s,
    }


# --------------------------------------------------------------------------
# bootstrap test  (bootstrap_permutation.py:467-766)
# --------------------------------------------------------------------------
def bootstrap_test(pls_alg, X, Y, U, s, V, cond_order, mctype, niter,
                   bscan=None, Xbscan=None, Ybscan=None, lvcorrs_orig=None,
                   Tvsc_orig=None, CI=0.95, sampler=None, keep_right=True, contrast=None):
    """``contrast``: the (normalised) contrast matrix of cst / csb / cmb
    (:658-675, :703).  For csb the reference's last step subtracts a
    (cells*b x q) array from the q x q ``lvcorrs_orig`` it is handed
    (pls_classes.py:1158) and raises a broadcasting ValueError; so does this."""
    sampler = sampler or Sampler()
    cond_order = np.asarray(cond_order)
    k = U.shape[1]
    multi = pls_alg in MULTI_ALGS
    right = np.empty((niter, V.shape[0], k))                   # :497
    left = None
    Tdist = None
    LVcorr = None
    bco = cond_order[:, bscan] if bscan is not None else cond_order    # csb: bscan is None (Q20)
    for i in range(niter):
        Yn = None
        for _ in range(100):                                   # :543-570
            if multi:
                ti = sampler.boot(cond_order)                  # :547
                bi = sampler.boot(cond_order[:, bscan])        # :551
                Xt, Xn, Yn = X[ti], Xbscan[bi], Ybscan[bi]
            else:
                bi = sampler.boot(cond_order)                  # :557
                Xn = X[bi]
                if Y is not None:
                    Yn = Y[bi]
            if Yn is None or not (group_stds(Yn, cond_order) == 0).any():   # :563 (Q8)
                break
        else:
            raise Exception("degenerate behaviour data")       # :572

        if pls_alg == "mct":
            permuted = mean_centre(Xn, cond_order, mctype)[1]  # :603
        elif pls_alg == "cst":
            permuted = group_condition_means(Xn, cond_order)   # :607
        elif pls_alg in BEHAV_ALGS:
            permuted = compute_corr(Xn, Yn, cond_order)        # :613
        else:
            permuted = create_multiblock(Xt, cond_order, pls_alg, bscan, mctype,
                                         Xbscan=Xn, Ybscan=Yn)   # :610

        U_hat = (V.T @ permuted.T).T                           # :617
        VS = permuted.T @ U                                    # :620
        V_hat = normalize(VS)                                  # :623
        right[i] = VS                                          # :626

        if pls_alg == "mct":                                   # :629-634
            if left is None:
                left = np.empty((niter,) + U_hat.shape)
                Tdist = np.empty((niter, cond_order.size, k))
            left[i] = U_hat
            Tdist[i] = group_condition_means(X @ V_hat, cond_order)
        elif pls_alg == "rb":                                  # :636-642
            lc = compute_corr(Xn @ V_hat, Yn, cond_order)
            if LVcorr is None:
                LVcorr = np.empty((niter,) + lc.shape)
            LVcorr[i] = lc
        elif pls_alg == "mb":                                  # :644-656
            lc = compute_corr(Xn @ V_hat, Yn, cond_order[:, bscan])
            if LVcorr is None:
                LVcorr = np.empty((niter,) + lc.shape)
                Tdist = np.empty((niter, cond_order.size, k))
            LVcorr[i] = lc
            sm = calculate_smeanmat(Xt, cond_order, mctype)
            Tdist[i] = group_condition_means(sm @ V_hat, cond_order)
        if contrast is not None:                               # :658-675
            cross = normalize(contrast).T @ permuted
            ncb = normalize(cross.T)
            if pls_alg in ("cmb", "cst"):
                if Tdist is None:
                    Tdist = np.empty((niter, cond_order.size, k))
                Tdist[i] = group_condition_means(X @ ncb, cond_order)   # :665-666
            if pls_alg in ("cmb", "csb"):
                lc = compute_corr(Xn @ ncb, Yn, bco)           # :671-674
                if LVcorr is None:
                    LVcorr = np.empty((niter,) + lc.shape)
                LVcorr[i] = lc

    std_errs = np.std(right, axis=0)                           # :695
    with np.errstate(divide="ignore", invalid="ignore"):
        boot_ratios = (V * s) / std_errs if contrast is None else V / std_errs   # :700-703
    z = _norm.ppf(1 - (1 - CI) / 2)                            # :709
    out = {"std_errs": std_errs, "boot_ratios": boot_ratios}
    if pls_alg in TASK_ALGS:                                   # :711-717
        half = np.std(Tdist, axis=0) * z
        out["conf_ints"] = (Tvsc_orig - half, Tvsc_orig + half)
        out["left_sv_sampled"] = left
        out["Tdistrib"] = Tdist
    else:                                                      # :723-734
        half = np.std(LVcorr, axis=0) * z
        out["conf_ints"] = (lvcorrs_orig - half, lvcorrs_orig + half)
        out["LVcorr"] = LVcorr
        out["left_sv_sampled"] = LVcorr
        if multi:
            half = np.std(Tdist, axis=0) * z
            out["conf_ints_T"] = (Tvsc_orig - half, Tvsc_orig + half)
            out["Tdistrib"] = Tdist
    if keep_right:
        out["right_sv_sampled"] = right
    return out


# --------------------------------------------------------------------------
# split-half  (split_half_resampling.py)
# --------------------------------------------------------------------------
def _cond_order_for(groups, nc):
    return np.array([[g] * nc for g in groups])


def _split_dim(pls_alg, p, cond_order, Y, bscan, Ybscan, contrasts=None):
    ng, nc = np.asarray(cond_order).shape
    if contrasts is not None:
        return min(p, contrasts.shape[1])                      # :84
    if pls_alg == "mct":
        return min(p, nc * ng)                                 # :80
    if pls_alg == "mb":
        return min(p, nc * ng + len(bscan) * ng * Ybscan.shape[1])   # :82
    return min(p, nc * ng * Y.shape[1])                        # :86


def _draw_split(sampler, tables, nc, bscan):
    """split_half_resampling.py:130-169 / :548-586."""
    i1, i2, b1, b2, g1, g2 = [], [], [], [], [], []
    for tbl in tables:
        ns = tbl.shape[0]
        half = int(np.floor(ns / 2))
        t = tbl[sampler.perm_rows(ns), :]                      # :136
        i1.append(t[:half, :].flatten())
        i2.append(t[half:, :].flatten())
        g1.append(len(i1[-1]) // nc)
        g2.append(len(i2[-1]) // nc)
        if bscan is not None:
            b1.append(t[:half][:, bscan].flatten())
            b2.append(t[half:][:, bscan].flatten())
    cat = np.concatenate
    return (cat(i1), cat(i2), (cat(b1), cat(b2)) if bscan is not None else None, g1, g2)


def _half_block(pls_alg, X1, Y1, co1, mctype, bscan, Xb1, Yb1):
    if pls_alg == "mct":
        return mean_centre(X1, co1, mctype)[1]                 # :186
    if pls_alg == "cst":
        return group_condition_means(X1, co1)                  # :212
    if pls_alg in BEHAV_ALGS:
        return compute_corr(X1, Y1, co1)                       # :203, :228
    return create_multiblock(X1, co1, pls_alg, bscan, mctype, Xbscan=Xb1, Ybscan=Yb1)   # :245


def split_half_both(pls_alg, matrix, Y, cond_order, num_split, mctype=None,
                    bscan=None, Ybscan=None, lv=1, CI=0.95, which="tt",
                    sampler=None, contrasts=None, only=None):
    """``which="tt"`` -> split_half_test_train (:23-401);
    ``which="sh"`` -> split_half (:404-861).  Same splitting code in both.

    ``only``: optional set of split numbers.  Every split is still drawn (the RNG stream is
    the reference's), but only the listed real and null splits are decomposed -- the other
    slabs stay zero and the summary statistics are then meaningless.  For full-size checks,
    where one split costs seconds of LAPACK time."""
    sampler = sampler or Sampler()
    cond_order = np.asarray(cond_order)
    n, p = matrix.shape
    ng, nc = cond_order.shape
    d = _split_dim(pls_alg, p, cond_order, Y, bscan, Ybscan, contrasts)

    def decomp(M):
        return run_pls(M) if contrasts is None else run_pls_contrast(M, contrasts)   # :194 / :216
    A = np.zeros((d, d, num_split))
    B = np.zeros((d, d, num_split))
    A0 = np.zeros((d, d, num_split))
    B0 = np.zeros((d, d, num_split))
    tables = subject_table(cond_order)
    alltab = np.concatenate(tables)
    multi = pls_alg in MULTI_ALGS
    behav = pls_alg in BEHAV_ALGS
    g1 = g2 = None

    def decompose(i, X1, X2, Y1, Y2, Xb1, Yb1, Xb2, Yb2, co1, co2, outA, outB):
        if only is not None and i not in only:
            return
        M1 = _half_block(pls_alg, X1, Y1, co1, mctype, bscan, Xb1, Yb1)
        M2 = _half_block(pls_alg, X2, Y2, co2, mctype, bscan, Xb2, Yb2)
        U1, s1, V1 = decomp(M1)
        if which == "tt":
            outA[:, :, i] = s1                                 # :195 (Q11 broadcast)
            outB[:, :, i] = V1.T @ M2.T @ U1                   # :196
        else:
            U2, _, V2 = decomp(M2)
            outA[:, :, i] = V1.T @ V2                          # :682
            outB[:, :, i] = U1.T @ U2                          # :683

    for i in range(num_split):
        i1, i2, bs, g1, g2 = _draw_split(sampler, tables, nc, bscan if multi else None)
        co1, co2 = _cond_order_for(g1, nc), _cond_order_for(g2, nc)
        X1, X2 = matrix[i1], matrix[i2]
        Y1 = Y2 = Xb1 = Xb2 = Yb1 = Yb2 = None
        if behav:
            Y1, Y2 = Y[i1], Y[i2]
        if multi:
            Xb1, Yb1 = matrix[bs[0]], Y[bs[0]]
            Xb2, Yb2 = matrix[bs[1]], Y[bs[1]]
        decompose(i, X1, X2, Y1, Y2, Xb1, Yb1, Xb2, Yb2, co1, co2, A, B)

    # null distribution (:264-383 / :685-802); reuses the LAST split's group
    # sizes (Q12)
    nsub = n // nc
    half = sum(g1)
    co1, co2 = _cond_order_for(g1, nc), _cond_order_for(g2, nc)
    for i in range(num_split):
        t = alltab[sampler.perm_rows(nsub), :]                 # :271
        i1, i2 = t[:half, :].flatten(), t[half:, :].flatten()
        if multi:
            b1, b2 = t[:half][:, bscan].flatten(), t[half:][:, bscan].flatten()
        if pls_alg in ("mct", "cst", "mb", "cmb"):
            permx = matrix[sampler.perm_rows(n)]               # :282
        else:
            permx = matrix
        X1, X2 = permx[i1], permx[i2]
        Y1 = Y2 = Xb1 = Xb2 = Yb1 = Yb2 = None
        if behav:
            permy = Y[sampler.perm_rows(n)]                    # :316, :340
            Y1, Y2 = permy[i1], permy[i2]
        if multi:
            Xb1, Yb1 = permx[b1], Y[b1]                        # :358
            Xb2, Yb2 = permx[b2], Y[b2]
        decompose(i, X1, X2, Y1, Y2, Xb1, Yb1, Xb2, Yb2, co1, co2, A0, B0)

    if which == "tt":
        return summarise_test_train(A, B, A0, B0)
    return summarise_split_half(A, B, A0, B0, lv, CI)


def summarise_test_train(train, test, train0, test0):
    """split_half_resampling.py:387-400."""
    d = train.shape[0]
    return {
        "pls_s_train": train,
        "pls_s_test": test,
        "z": [np.mean(test[i, i, :]) / np.std(test[i, i, :], ddof=1) for i in range(d)],
        "pls_s_train_null": train0,
        "pls_s_test_null": test0,
        "z_null": [np.mean(test0[i, i, :]) / np.std(test0[i, i, :], ddof=1) for i in range(d)],
    }


def summarise_split_half(u, v, u0, v0, lv, CI):
    """split_half_resampling.py:805-859, including passing CI (0..1) straight
    to np.percentile (Q10)."""
    out = {}

    def stats(prefix, arr, tag, with_std):
        diag = [np.abs(arr[i, i, :]) for i in range(lv)]
        out[f"{prefix}_mean_{tag}"] = [np.mean(x) for x in diag]
        if with_std:
            out[f"{prefix}_std_{tag}"] = [np.std(x) for x in diag]
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


# ---------------------------------------------------------------------------
# upstream feed (plspy/io/io.py) -- the reference module imports nibabel at its top and cannot be
# imported here: these restatements are NOT pinned by reference-generated fixtures ("parity
# unpinned"); they are checked against the reference's own round-trip property (tests/test_io.py:8-36)
# ---------------------------------------------------------------------------
def io_create_threshold_mask_from_matrices(matrices, threshold=0.15):
    """io.py:353-398: True where the mean over subjects of the time means exceeds
    threshold * (max - min) + min."""
    if threshold < 0 or threshold > 1:
        raise ValueError(f"threshold must be greater than 0 or less than 1. Value passed in : {threshold}")
    mats = np.array(matrices)
    mean_all = np.mean(np.mean(mats, axis=1), axis=0)
    cond = mean_all > (threshold * (np.max(mean_all) - np.min(mean_all)) + np.min(mean_all))
    return np.ma.masked_where(cond, mean_all).mask


def io_apply_mask_matrices(matrices, mask):
    """io.py:427-460: m[np.broadcast_to(mask, m.shape)] for every matrix."""
    return [m[np.broadcast_to(mask, m.shape)] for m in matrices]


def io_concat_flatten_all_groups(groups_list):
    """io.py:680-698."""
    full = np.concatenate(groups_list, axis=0)
    return full.reshape(full.shape[0], -1)


def io_remap_vectorized_subject_to_4d(vector, mask, original_shape):
    """io.py:701-760: the masked vector back in its (time, x, y, z) volume, zeros elsewhere."""
    out = np.zeros(original_shape)
    out[:, mask == True] = vector.reshape(original_shape[0], -1)      # noqa: E712
    return out
