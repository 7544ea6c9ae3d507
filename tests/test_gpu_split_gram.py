"""K2s (plsr_split_gram): the two-stage per-split Gram of behaviour / multiblock PLS against a direct
NumPy statement of the stacked cross-blocks (class_functions.py:185-247, :454-516 as the oracle restates
them) and against the round-2 fused Gram (plsr_gram_fused) on the same items."""
import numpy as np
import pytest

from _util import assert_close

pytestmark = pytest.mark.gpu


def _zs(M):
    """Per-column z-score of a cell's rows (ddof 0) / sqrt(rows), constant columns -> 0."""
    mu = M.mean(0)
    sd = np.sqrt(((M - mu) ** 2).mean(0))
    with np.errstate(divide="ignore", invalid="ignore"):
        z = (M - mu) / sd / np.sqrt(M.shape[0])
    z[:, ~(sd > np.finfo(float).eps * np.abs(mu))] = 0.0
    return z


def _numpy_gram(X, Y, cells, item):
    """Stacked cross-block of one item from the cell description, then its (normalised) Gram."""
    lo = np.concatenate(([0], np.cumsum(cells["cell_rows"])))
    xs, ys = cells["xsrc"][item], cells["ysrc"][item]
    nbq, b = cells["nbq"], Y.shape[1]
    beh = []
    sums = []
    for q, (a, e) in enumerate(zip(lo[:-1], lo[1:])):
        Xc = X[xs[a:e]]
        if q < nbq:
            beh.append(_zs(Y[ys[a:e]]).T @ _zs(Xc))                # b x p
        sums.append(Xc.sum(0))
    task = cells["Wc"] @ np.array(sums) if cells["Wc"] is not None else None
    rows = []
    for rc, rsub in zip(cells["row_cell"], cells["row_sub"]):
        rows.append(beh[rc][rsub] if rc >= 0 else task[rsub])
    M = np.array(rows)
    if cells["normalise"]:
        nrm = np.linalg.norm(M, axis=1)
        M = np.where(nrm[:, None] > 0, M / np.where(nrm > 0, nrm, 1.0)[:, None], 0.0)
    return M @ M.T


def _run(alg, groups, nc, b, bscan, p, S, seed, mctype=0, shift=0.0):
    from plspy_amd import split_half_resampling as sh
    from plspy_amd.engine import ProjectionEngine
    rs = np.random.RandomState(seed)
    n = sum(groups) * nc
    X = rs.randn(n, p) * (1 + rs.rand(1, p)) + shift * rs.randn(1, p)
    X[:, 5] = 3.25                                        # a constant voxel: z-scored cells give 0
    Y = rs.randn(n, b)
    co = np.array([[g] * nc for g in groups])
    np.random.seed(seed)
    eng = ProjectionEngine(X)
    _, item = sh._prepare(alg, X, Y, co, S, mctype, bscan, eng)
    cells = item["cells"]
    assert cells is not None
    res = eng.split_gram(cells, Y)
    return X, Y, cells, item, (res[0] if res is not None else None), eng


@pytest.mark.parametrize("case", [
    # alg, groups, conditions, behaviours, bscan, p, splits
    ("mb", (20, 20), 3, 8, [1, 2], 4103, 3),          # config 4's cell structure (exact instance), ragged last tile
    ("rb", (20, 20), 3, 8, None, 2048, 3),            # config 3's cell structure (exact instance)
    ("mb", (6, 5), 3, 2, [1, 2], 333, 4),             # small ragged cells, generic instance
    ("rb", (6, 6), 2, 2, None, 257, 4),
    ("rb", (7,), 3, 5, None, 130, 3),                 # one group, odd number of cells per half
    ("mb", (30,), 3, 3, [0, 2], 1000, 2),             # cells of 15 rows (five k-steps)
    ("cmb", (8, 9), 2, 4, [1], 500, 3),               # plain cell means as task rows
    ("rb", (36, 30), 2, 8, None, 700, 2),             # cells of 18 / 15 rows
])
def test_split_gram_matches_numpy(case):
    alg, groups, nc, b, bscan, p, S = case
    X, Y, cells, item, G, eng = _run(alg, groups, nc, b, bscan, p, S, seed=len(groups) + nc + b + p)
    assert G is not None, "the two-stage kernel declined a shape it is meant to serve"
    G = G.cpu().numpy()
    m = 2 * item["k"]
    for it in range(G.shape[0]):
        want = _numpy_gram(X, Y, cells, it)
        scale = np.abs(want).max()
        assert_close(G[it, :m, :m], want, 1e-10, 1e-12 * scale, f"{alg} item {it}")
        assert not G[it, m:].any() and not G[it, :, m:].any()


def test_split_gram_matches_fused_dense_path():
    """Same items through gram_phase's fused kernel (dense stacked operators, gather table)."""
    from plspy_amd import split_half_resampling as sh
    X, Y, cells, item, G, eng = _run("mb", (10, 12), 3, 4, [0, 1], 1500, 6, seed=11)
    assert G is not None
    dense = dict(item, cells=None)
    Gd = sh._grams(eng, dense, np.arange(item["S"]))[0].cpu().numpy()
    m = 2 * item["k"]
    assert_close(G.cpu().numpy()[:, :m, :m], Gd[:, :m, :m], 1e-10, 1e-12, "two-stage vs fused dense")


def test_split_gram_large_mean_voxels():
    """Voxel means 1e4 standard deviations away from zero: the two-pass cell statistics keep the
    correlation rows exact where a one-pass variance would lose eight digits."""
    X, Y, cells, item, G, eng = _run("rb", (20, 20), 3, 8, None, 640, 2, seed=5, shift=1e4)
    G = G.cpu().numpy()
    m = 2 * item["k"]
    for it in range(G.shape[0]):
        want = _numpy_gram(X, Y, cells, it)
        assert_close(G[it, :m, :m], want, 1e-9, 1e-11 * np.abs(want).max(), f"item {it}")


def test_split_gram_declines_what_it_cannot_serve():
    from plspy_amd import split_half_resampling as sh
    from plspy_amd.engine import ProjectionEngine
    rs = np.random.RandomState(0)
    groups, nc, b = (10, 10), 2, 9                       # nine behaviours: more than a half tile
    n = sum(groups) * nc
    X, Y = rs.randn(n, 300), rs.randn(n, b)
    co = np.array([[g] * nc for g in groups])
    eng = ProjectionEngine(X)
    np.random.seed(1)
    _, item = sh._prepare("rb", X, Y, co, 2, None, None, eng)
    assert eng.split_gram(item["cells"], Y) is None
    G, _ = sh._grams(eng, item, np.arange(item["S"]))     # ... and the caller falls back to the fused Gram
    assert G.shape[0] == item["S"]


def _numpy_rows(X, Y, cells, item):
    lo = np.concatenate(([0], np.cumsum(cells["cell_rows"])))
    xs, ys = cells["xsrc"][item], cells["ysrc"][item]
    beh, sums = [], []
    for q, (a, e) in enumerate(zip(lo[:-1], lo[1:])):
        Xc = X[xs[a:e]]
        if q < cells["nbq"]:
            beh.append(_zs(Y[ys[a:e]]).T @ _zs(Xc))
        sums.append(Xc.sum(0))
    task = cells["Wc"] @ np.array(sums) if cells["Wc"] is not None else None
    return np.array([beh[rc][rs] if rc >= 0 else task[rs] for rc, rs in zip(cells["row_cell"], cells["row_sub"])])


@pytest.mark.parametrize("case", [
    # groups, conditions, behaviours, bscan, p, items
    ((20, 20), 3, 8, [1, 2], 4103, 5),               # config 6's cell structure: the exact instance
    ((6, 5), 3, 2, [0, 2], 333, 4),                  # ragged cells
    ((9,), 2, 3, [1], 130, 3),                       # one group
])
def test_split_rows_of_bootstrap_samples(case):
    """The ROWS variant (plsr_split_rows) on bootstrap-shaped items -- rows drawn WITH replacement inside their
    cells, task and behaviour blocks drawn independently (bootstrap_permutation.py:547-553) -- against the
    direct NumPy statement of the un-normalised multiblock rows and their norms; then plsr_rows_project on
    top (normalise, project on U, moments)."""
    import torch
    from plspy_amd import class_functions as cf, operators
    from plspy_amd.engine import ProjectionEngine
    groups, nc, b, bscan, p, items = case
    rs = np.random.RandomState(p + items)
    co = np.array([[g] * nc for g in groups])
    n = int(co.sum())
    X = rs.randn(n, p) * (1 + rs.rand(1, p)) + rs.randn(1, p)
    Yb_rows = np.flatnonzero(cf.bscan_mask(co, bscan))
    Yb = rs.randn(len(Yb_rows), b)
    bt, bb = cf.cell_bounds(co), cf.cell_bounds(co[:, bscan])
    # with-replacement draws inside every cell (the reference draws subjects; any multiset will do here)
    ti = np.concatenate([rs.randint(lo, hi, size=(items, hi - lo)) for lo, hi in zip(bt[:-1], bt[1:])], axis=1)
    bi = np.concatenate([rs.randint(lo, hi, size=(items, hi - lo)) for lo, hi in zip(bb[:-1], bb[1:])], axis=1)
    W = operators.mean_centre_operator(co, 0)
    Wcell = W[:, bt[:-1]]
    ng, nbs = len(groups), len(bscan)
    per = nc + nbs * b
    row_cell, row_sub = [], []
    for g in range(ng):
        for r in range(per):
            row_cell.append(-1 if r < nc else g * nbs + (r - nc) // b)
            row_sub.append(g * nc + r if r < nc else (r - nc) % b)
    ncb = len(bb) - 1
    cells = dict(xsrc=np.concatenate((Yb_rows[bi], ti), axis=1), ysrc=np.concatenate((bi, np.zeros_like(ti)), axis=1),
                 cell_rows=[int(x) for x in np.diff(bb)] + [int(x) for x in np.diff(bt)], nbq=ncb,
                 Wc=np.concatenate((np.zeros((ng * nc, ncb)), Wcell), axis=1), row_cell=row_cell, row_sub=row_sub)
    eng = ProjectionEngine(X)
    got = eng.split_rows(cells, Yb)
    assert got is not None
    R, rowsq = got
    kr = ng * per
    want = np.stack([_numpy_rows(X, Yb, cells, i) for i in range(items)])
    scale = np.abs(want).max()
    np.testing.assert_allclose(R.cpu().numpy(), want, rtol=1e-10, atol=1e-12 * scale)
    np.testing.assert_allclose(rowsq.cpu().numpy()[:, :kr], (want ** 2).sum(-1), rtol=1e-10, atol=1e-20)
    # second pass on top: VS = (U^T D^-1) R, shifted moments
    U = np.linalg.qr(rs.randn(kr, kr))[0]
    ref = rs.randn(p, kr)
    S1 = torch.zeros((p, kr), dtype=torch.float64, device=eng.device)
    S2 = torch.zeros_like(S1)
    assert eng.rows_project(R, rowsq, U, ref=ref, S1=S1, S2=S2)
    nrm = np.sqrt((want ** 2).sum(-1))
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = np.where(nrm > 0, 1.0 / nrm, 0.0)
    VS = np.einsum("rj,br,brv->bjv", U, inv, want)
    np.testing.assert_allclose(R.cpu().numpy(), VS, rtol=1e-9, atol=1e-11 * np.abs(VS).max())
    d = np.transpose(VS, (0, 2, 1)) - ref
    np.testing.assert_allclose(S2.cpu().numpy(), (d ** 2).sum(0), rtol=1e-9, atol=1e-10 * items)


def test_split_gram_random_cell_structures():
    """Random cell structures straight through engine.split_gram / split_rows (not through the split-half
    drawing code): one to twenty cells of 1..20 rows, 1..8 behaviours, with and without task rows, rows
    repeated inside cells, voxel counts off the tile -- every guarded instance, against NumPy."""
    from plspy_amd.engine import ProjectionEngine
    rs = np.random.RandomState(2024)
    served = 0
    for trial in range(40):
        n = int(rs.randint(8, 90))
        p = int(rs.choice([16, 17, 130, 257, 1000]))
        b = int(rs.randint(1, 9))
        rmax = int(rs.choice([4, 12, 20]))
        nbq = int(rs.randint(1, 7 if rmax > 12 else 11))
        ntask = int(rs.randint(0, min(8, 20 - nbq) + 1))
        ktask = int(rs.randint(1, 9)) if ntask else 0
        cell_rows = [int(rs.randint(1, rmax + 1)) for _ in range(nbq + ntask)]
        nz = sum(cell_rows)
        items = int(rs.randint(1, 5))
        X = rs.randn(n, p) * 2 + rs.randn(1, p)
        X[:, 3] = -1.5                                         # a constant voxel
        Y = rs.randn(n, b)
        if b > 1:
            Y[:, 1] = 0.25                                     # a constant behaviour
        Wc = np.concatenate((np.zeros((ktask, nbq)), rs.randn(ktask, ntask)), axis=1) if ntask else None
        row_cell = [q for q in range(nbq) for _ in range(b)] + [-1] * ktask
        row_sub = [s for _ in range(nbq) for s in range(b)] + list(range(ktask))
        perm = rs.permutation(len(row_cell))                   # any logical order of the rows
        cells = dict(xsrc=rs.randint(0, n, size=(items, nz)), ysrc=rs.randint(0, n, size=(items, nz)),
                     cell_rows=cell_rows, nbq=nbq, Wc=Wc, normalise=bool(trial % 2),
                     row_cell=[row_cell[i] for i in perm], row_sub=[row_sub[i] for i in perm])
        eng = ProjectionEngine(X)
        tag = f"trial {trial}: n={n} p={p} b={b} cells={cell_rows} nbq={nbq} ktask={ktask}"
        res = eng.split_gram(cells, Y)
        rows = eng.split_rows(cells, Y)
        if res is None:
            assert rows is None or len(row_cell) <= 96, tag
            continue
        served += 1
        m = len(row_cell)
        G = res[0].cpu().numpy()
        for it in range(items):
            want = _numpy_gram(X, Y, cells, it)
            assert_close(G[it, :m, :m], want, 1e-9, 1e-11 * max(np.abs(want).max(), 1e-300), tag)
        if rows is not None:
            R, rowsq = rows
            wantR = np.stack([_numpy_rows(X, Y, cells, i) for i in range(items)])
            np.testing.assert_allclose(R.cpu().numpy(), wantR, rtol=1e-9, atol=1e-11 * max(np.abs(wantR).max(), 1e-300),
                                       err_msg=tag)
            np.testing.assert_allclose(rowsq.cpu().numpy()[:, :m], (wantR ** 2).sum(-1), rtol=1e-9, atol=1e-18, err_msg=tag)
    assert served >= 25, served
