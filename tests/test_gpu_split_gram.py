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
