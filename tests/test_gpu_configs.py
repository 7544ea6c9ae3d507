"""GPU parity at BASELINE.json's FULL shapes (configs 2-5), i.e. the launch shapes and kernel
instances the benchmarks time -- which the golden fixtures (p <= 1000, R <= 100) never reach.

The oracle's direct form cannot run a whole phase at these sizes, so each
statistic is compared through something that can:

* voxel-LOCAL statistics (right_sv_sampled -> std_errs, boot_ratios, moment sums): the oracle /
  direct NumPy on a voxel subset of X, over ALL resamples of the phase;
* p-wide contractions of the linear (task) path (s_hat^2, Tdistrib numerators): the Gram
  identity a^T (X X^T) a -- p-free, computed by NumPy, independent of the
  kernels' tiling and reduction order;
* p-wide nonlinear results (rb LVcorr, mb split-half slabs): the oracle's direct form at full
  size for a handful of resamples / splits, chosen to straddle batch boundaries.

Reference arithmetic: bootstrap_permutation.py:404-405, :617-634, :695; split_half_resampling.py
:186-196, :612-683; class_functions.py:185-247, :454-516.  Tolerances are those of BASELINE.md
(s_hat 1e-10 rel, bootstrap statistics 1e-9 / 1e-8 rel)."""
import ctypes
import warnings

import numpy as np
import pytest

from oracle import plspy_oracle as orc
from tests._util import assert_close

pytestmark = pytest.mark.gpu


def _operators(M, inds):
    """a[b] = P_b^T M (R, n, k): the direct statement of `W @ X[inds_b]` projected on U as an
    operator on X -- VS_b = X^T a[b] with M = W^T U."""
    R, n = inds.shape
    a = np.zeros((R, n, M.shape[1]))
    np.add.at(a, (np.arange(R)[:, None], inds), M[None])      # row i of X[inds] is X[inds[i]]
    return a


def _mct_problem(groups, nc, p, seed=0):
    from plspy_amd import operators
    co = np.array([[g] * nc for g in groups])
    X = np.random.RandomState(seed).randn(int(co.sum()), p)
    W = operators.mean_centre_operator(co, 0)
    Wm = operators.cell_mean_operator(co)
    U, s, Vt = np.linalg.svd(W @ X, full_matrices=False)
    s[np.abs(s) < 1e-12] = 0
    return co, X, W, Wm, U, s, Vt.T


def _check_mct_phases(X, co, W, Wm, U, s, V, nperm, nboot, sub, overlap_tail=True):
    """One bench.py step at the given shape, every output against NumPy."""
    import torch
    from plspy_amd import resample
    from plspy_amd.engine import ProjectionEngine
    n, p = X.shape
    k = U.shape[1]
    eng = ProjectionEngine(X)
    M = W.T @ U
    ref = V * s
    Xm = eng.apply_operator(Wm)
    np.random.seed(1234)
    pinds = resample.task_permutations(co, nperm)
    binds = resample.bootstraps(co, nboot)
    res = eng.boot_phase(k, inds=eng.dev(binds, torch.int32), M=eng.dev(M), ref=eng.dev(ref), Xm=Xm,
                         overlap_tail=overlap_tail)
    ssq_p = eng.perm_phase(k, inds=eng.dev(pinds, torch.int32), M=eng.dev(M))
    eng.join()
    sd, ratio = eng.boot_finalize(res["S1"], res["S2"], nboot, num=eng.dev(ref))
    torch.cuda.synchronize()
    ssq_p, ssq_b, T, S1, S2, sd, ratio = (t.cpu().numpy() for t in
                                          (ssq_p, res["ssq"], res["T"], res["S1"], res["S2"], sd, ratio))
    live = s > 1e-10 * s.max()
    G = X @ X.T
    # permutation: s_hat^2[b, j] = a_bj^T G a_bj   (bootstrap_permutation.py:404-405)
    a = _operators(M, pinds)
    want = np.einsum("bij,il,blj->bj", a, G, a)
    assert_close(ssq_p[:, live], want[:, live], 1e-10, 0, "perm s_hat^2 (Gram identity)")
    assert np.all(np.abs(ssq_p[:, ~live]) <= 1e-18 * want.max())
    # bootstrap: column norms (:623) and Tdistrib numerators (:633-634) the same way
    a = _operators(M, binds)
    want = np.einsum("bij,il,blj->bj", a, G, a)
    assert_close(ssq_b[:, live], want[:, live], 1e-10, 0, "boot column norms^2 (Gram identity)")
    wantT = np.einsum("ci,il,blj->bjc", Wm, G, a)
    assert_close(T[:, live], wantT[:, live], 1e-9, 1e-11 * np.abs(wantT).max(), "Tdistrib numerators")
    # voxel-local: every bootstrap's projection of a voxel subset, directly (:617-626, :695)
    VS = np.einsum("iv,bij->bvj", X[:, sub], a)                  # (R, |sub|, k) = right_sv_sampled[:, sub]
    d = VS - ref[sub]
    scale = np.abs(VS).max()
    assert_close(S1[sub], d.sum(0), 1e-9, 1e-11 * scale * nboot, "S1 (shifted first moment)")
    assert_close(S2[sub], (d ** 2).sum(0), 1e-9, 1e-11 * scale ** 2 * nboot, "S2 (shifted second moment)")
    assert_close(sd[sub][:, live], np.std(VS, axis=0)[:, live], 1e-9, 1e-13, "std_errs")
    assert_close(ratio[sub][:, live], (ref[sub] / np.std(VS, axis=0))[:, live], 1e-8, 1e-10, "boot_ratios")
    return eng


def _subset(p, m=500, seed=3):
    sub = np.random.RandomState(seed).choice(p, m, replace=False)
    return np.unique(np.concatenate((sub, [0, 63, 64, 65, p // 2, p - 65, p - 64, p - 1])))


def test_config2_bench_launch_shape_msplit_ge_2():
    """BASELINE config 2 exactly as bench.py launches it: 60 x 200 000, 1000 + 1000 resamples in
    one launch each.  At this shape the bootstrap kernel (K1br) runs two runs per latent
    variable over 63 batch tiles (msplit = 2, an odd tile count): second-moment partials per
    run, first moment from the summed operator by the first run only."""
    co, X, W, Wm, U, s, V = _mct_problem((10, 10), 3, 200_000)
    eng = _check_mct_phases(X, co, W, Wm, U, s, V, 1000, 1000, _subset(200_000))
    plan = eng.plan(6, 1000, k2=6, boot=True)
    assert plan["register_resident"] and plan["splits"] >= 2, plan      # the msplit >= 2 branch ran
    assert plan["tiles"] == 63 and plan["tiles"] % plan["splits"] != 0, plan
    assert eng.batch_size(6, 6, 1000) == 1000                           # one launch, like the bench
    assert eng.plan(6, 1000, boot=False)["splits"] > 1


@pytest.mark.parametrize("p,R", [(3001, 272), (3001, 1000), (130, 2000)])
def test_k1br_splits_small_p(p, R):
    """msplit up to tpl / 4 with few voxel tiles (want is large): tile counts that do not divide
    by the number of runs, compared in full with direct NumPy."""
    co, X, W, Wm, U, s, V = _mct_problem((10, 10), 3, p, seed=p)
    eng = _check_mct_phases(X, co, W, Wm, U, s, V, R // 2 + 3, R, np.arange(p), overlap_tail=False)
    plan = eng.plan(6, R, k2=6, boot=True)
    assert plan["register_resident"] and plan["splits"] >= 4, plan
    assert plan["tiles"] % plan["splits"] != 0, plan


def test_config5_shape_lds_fed_kernels():
    """BASELINE config 5's shape (mct 240 x 500 000, k = 12): the LDS-fed projection kernel with one
    workgroup of ten (bootstrap, period 3, NH 3) / fourteen (permutation) waves per CU."""
    co, X, W, Wm, U, s, V = _mct_problem((20, 20, 20, 20), 3, 500_000)
    eng = _check_mct_phases(X, co, W, Wm, U, s, V, 112, 104, _subset(500_000, 300))
    plan = eng.plan(12, 104, k2=12, boot=True)
    assert not plan["register_resident"] and plan["voxel_tiles"] == 7813, plan


def test_mct_k16_n128_no_spill_shape():
    """An mct shape that used to run a spilling instance (LDS-fed bootstrap kernel, period 4:
    k = 16, n = 128)."""
    co, X, W, Wm, U, s, V = _mct_problem((8,) * 8, 2, 20_000, seed=8)
    _check_mct_phases(X, co, W, Wm, U, s, V, 40, 72, _subset(20_000, 300))


# ---------------------------------------------------------------------------
# config 3: behaviour PLS 120 x 200 000, Y 120 x 8 (k = 48)
# ---------------------------------------------------------------------------
@pytest.fixture(scope="module")
def rb_problem():
    co = np.array([[20] * 3, [20] * 3])
    X = np.random.RandomState(0).randn(120, 200_000)
    Y = np.random.RandomState(1).randn(120, 8)
    obs = orc.observed("rb", X, co, Y=Y)
    return co, X, Y, obs


def test_config3_rb_bootstrap_full_batch_plus_ragged(rb_problem):
    """135 bootstraps = one full batch of 125 items (24 GiB scratch limit) + a ragged one of 10,
    through K4b (item_beh_kernel, tile-major VS^T) and K5i (latent_wave_kernel on each sample's different rows)."""
    from plspy_amd.bootstrap_permutation import ResampleTest
    co, X, Y, obs = rb_problem
    U, s, V = obs["U"], obs["s"], obs["V"]
    nboot = 135
    np.random.seed(77)
    rt = ResampleTest._create("rb", X, Y, U, s.copy(), V, co, None, nperm=0, nboot=nboot,
                              lvcorrs_orig=obs["lvcorrs"])
    inds = rt.boot_debug_dict["indices"]
    assert inds.shape == (nboot, 120)
    # voxel-local statistics over all 135 bootstraps: the oracle on a voxel subset, same draws
    sub = _subset(200_000, 250)
    with warnings.catch_warnings(), np.errstate(all="ignore"):
        warnings.simplefilter("ignore")
        ref = orc.bootstrap_test("rb", X[:, sub], Y, U, s, V[sub], co, None, nboot, lvcorrs_orig=obs["lvcorrs"],
                                 sampler=orc.ReplaySampler(list(inds)), keep_right=False)
    assert_close(rt.std_errs[sub], ref["std_errs"], 1e-8, 1e-13, "std_errs (voxel subset, all bootstraps)")
    assert_close(rt.boot_ratios[sub], ref["boot_ratios"], 1e-8, 1e-10, "boot_ratios")
    assert np.isfinite(rt.std_errs).all() and (rt.std_errs > 0).all()
    # p-wide: LVcorr of single bootstraps through the oracle's direct form at full size
    # (first / last of the full batch, first / last of the ragged one)
    for b in (0, 124, 125, 134):
        Xn, Yn = X[inds[b]], Y[inds[b]]
        VS = orc.compute_corr(Xn, Yn, co).T @ U                     # :613, :620
        lc = orc.compute_corr(Xn @ orc.normalize(VS), Yn, co)       # :623, :638-641
        assert_close(rt.LVcorr[b], lc, 1e-8, 1e-11, f"LVcorr[{b}]")
    z = 1.959963984540054
    half = np.std(rt.LVcorr, axis=0) * z
    assert_close(rt.conf_ints[0], obs["lvcorrs"] - half, 1e-12, 1e-14, "conf lo")


def test_config3_rb_permutation(rb_problem):
    """rb permutation phase at full size: project_kernel<1,0,0> with eight waves on the z-scored X
    (n = 120).  Selected permutations against the direct form; counts against those values."""
    from plspy_amd.bootstrap_permutation import ResampleTest
    co, X, Y, obs = rb_problem
    U, s, V = obs["U"], obs["s"], obs["V"]
    nperm = 530                                   # two device batches (512 + 18)
    np.random.seed(78)
    rt = ResampleTest._create("rb", X, Y, U, s.copy(), V, co, None, nperm=nperm, nboot=0)
    perms = rt.perm_debug_dict["indices"]
    s_list = rt.perm_debug_dict["s_list"]
    assert s_list.shape == (nperm, 48) and np.isfinite(s_list).all()
    for b in (0, 511, 512, 529):
        VS = orc.compute_corr(X, Y[perms[b]], co).T @ U             # :338, :396, :404
        assert_close(s_list[b], np.sqrt((VS ** 2).sum(0)), 1e-10, 0, f"s_hat[{b}]")
    # sum_j s_hat_j^2 = ||R_b||_F^2 is p-free given the z-scored X: check every permutation
    bounds = np.concatenate(([0], np.cumsum(co.reshape(-1))))
    Xz = np.concatenate([np.nan_to_num(orc._zscore_cell(X[lo:hi])) / np.sqrt(hi - lo)
                         for lo, hi in zip(bounds[:-1], bounds[1:])])
    tot = np.zeros(nperm)
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        Gc = Xz[lo:hi] @ Xz[lo:hi].T
        Yp = Y[perms[:, lo:hi]]                                      # (R, n_c, b)
        Yz = (Yp - Yp.mean(1, keepdims=True)) / Yp.std(1, keepdims=True) / np.sqrt(hi - lo)
        tot += np.einsum("rib,ij,rjb->r", Yz, Gc, Yz)
    assert_close((s_list ** 2).sum(1), tot, 1e-10, 0, "sum of s_hat^2 per permutation")
    np.testing.assert_array_equal(rt.permute_ratio, (s_list >= s).sum(0) / (nperm + 1))


def _align_signs(ours, ref):
    """Singular vectors come with arbitrary signs (LAPACK's in the oracle, Jacobi's here), so entry (i, j) of
    a split-half slab (V1.T M2.T U1, V1.T V2, U1.T U2) may differ from the oracle's by s_i t_j with two
    vectors of +-1.  Finds them (alternating majority votes weighted by the entries' size) and returns ours
    with the oracle's signs."""
    w = ours * ref
    t = np.where(w[np.argmax(np.abs(w).sum(1))] < 0, -1.0, 1.0)      # the signs along the heaviest row
    for _ in range(6):
        s = np.where(w @ t < 0, -1.0, 1.0)
        t = np.where(s @ w < 0, -1.0, 1.0)
    return ours * s[:, None] * t[None, :]


def _check_split_half_against_oracle(X, Y, S, only, live, s_tol, seed):
    """Config 4's multiblock split-half (tt + sh, real + null) against the oracle's LAPACK path on the splits
    in `only`: EVERY structurally live singular value to s_tol, the full live x live slabs after sign
    alignment; the two latent variables that are null for any data (rank-deficient task block) are exact
    zeros here and rounding noise in the oracle."""
    from plspy_amd import split_half_resampling as sh
    co = np.array([[20] * 3, [20] * 3])
    bscan, lv = [1, 2], 2
    mask = orc.bscan_mask(co, bscan)
    kw = dict(mctype=0, bscan=bscan, Xbscan=X[mask], Ybscan=Y[mask])
    np.random.seed(seed)
    tt = sh.split_half_test_train("mb", X, Y, co, S, **kw)
    np.random.seed(seed)
    res = sh.split_half("mb", X, Y, co, S, lv=lv, CI=0.95, **kw)
    okw = dict(mctype=0, bscan=bscan, Ybscan=Y[mask], lv=lv, only=only)
    with warnings.catch_warnings(), np.errstate(all="ignore"):
        warnings.simplefilter("ignore")
        np.random.seed(seed)
        ott = orc.split_half_both("mb", X, Y, co, S, which="tt", **okw)
        np.random.seed(seed)
        osh = orc.split_half_both("mb", X, Y, co, S, which="sh", **okw)
    assert tt["pls_s_train"].shape == (38, 38, S)
    ratios = []
    for i in sorted(only):
        for key in ("pls_s_train", "pls_s_train_null"):
            got, want = tt[key][0, :, i], ott[key][0, :, i]
            assert_close(got[:live], want[:live], s_tol, 0, f"{key}[{i}]")
            assert not got[live:].any() and (want[live:] < 1e-11 * want[0]).all(), f"{key}[{i}] null latent variables"
            ratios.append(want[live - 1] / want[0])
        for key, mine, theirs in (("pls_s_test", tt, ott), ("pls_s_test_null", tt, ott),
                                  ("pls_dist_u", res, osh), ("pls_dist_v", res, osh),
                                  ("pls_dist_null_u", res, osh), ("pls_dist_null_v", res, osh)):
            want = theirs[key][:live, :live, i]
            got = _align_signs(mine[key][:live, :live, i], want)
            assert_close(got, want, 1e-6, 1e-8 * np.abs(want).max(), f"{key}[{i}]")
    for key, val in tt.items():
        assert np.isfinite(np.asarray(val, dtype=float)[:live]).all(), key
    return ratios


# ---------------------------------------------------------------------------
# config 4: multiblock split-half, k = 38, two-stage Gram split_gram_kernel<4, 6, 1, 3, true>
# ---------------------------------------------------------------------------
def test_config4_mb_split_half_full_size():
    X = np.random.RandomState(0).randn(120, 200_000)
    Y = np.random.RandomState(1).randn(120, 8)
    S = 12
    _check_split_half_against_oracle(X, Y, S, {0, S - 1}, live=36, s_tol=1e-10, seed=41)


def test_config4_planted_graded_spectrum():
    """Behaviour columns that are nearly collinear (one common score + perturbations graded from 1 down to
    3e-4) make every half's cross-block graded: its singular values span four decades.  The eigenvalues of a
    Gram lose eps (s_1 / s_i)^2 -- 1e-8 at the small end -- so the split-half decomposition must refine
    these items (split_half_resampling._refine) to keep numpy.linalg.svd's accuracy on EVERY live singular
    value."""
    rs = np.random.RandomState(7)
    X = rs.randn(120, 200_000)
    Y = rs.randn(120, 1) + rs.randn(120, 8) * np.logspace(0, -3.5, 8)[None, :]
    S = 6
    ratios = _check_split_half_against_oracle(X, Y, S, {0, S - 1}, live=36, s_tol=1e-10, seed=3)
    assert max(ratios) < 1e-3 and min(ratios) > 1e-6, ratios         # graded, and within the planted range


def test_config6_mb_permutation_full_size():
    """Multiblock permutation at config 4's data through the per-resample Grams (K2s, exact instance <2,3,1,5>):
    s_list of selected permutations against the oracle's direct form at full size (its own multiblock of the permuted
    task rows / permuted behaviour, `.T @ U`, the Q3 rescaling), the ratios against the counts over those values."""
    from plspy_amd.bootstrap_permutation import ResampleTest
    from plspy_amd.engine import ProjectionEngine
    co = np.array([[20] * 3, [20] * 3])
    X = np.random.RandomState(0).randn(120, 200_000)
    Y = np.random.RandomState(1).randn(120, 8)
    bscan = [1, 2]
    obs = orc.observed("mb", X, co, Y=Y, mctype=0, bscan=bscan)
    U, s, V = obs["U"], obs["s"], obs["V"]
    nperm = 37
    np.random.seed(11)
    eng = ProjectionEngine(X)
    calls = []
    orig = eng.split_gram
    eng.split_gram = lambda cells, Yb: calls.append(len(cells["xsrc"])) or orig(cells, Yb)
    rt = ResampleTest._create("mb", X, Y, U, s.copy(), V, co, 0, nperm=nperm, nboot=0, bscan=bscan,
                              Xbscan=obs["Xbscan"], Ybscan=obs["Ybscan"], lvcorrs_orig=obs["lvcorrs"],
                              Tvsc_orig=obs["Tvsc_orig"], engine=eng)
    assert calls == [1, nperm]                        # (the observed block's total variance, then the permutations)
    draws = rt.perm_debug_dict["indices"]
    s_list = rt.perm_debug_dict["s_list"]
    assert s_list.shape == (nperm, 38) and np.isfinite(s_list).all()
    n = 120
    live = s > 1e-8 * s[0]
    for r in (0, 17, nperm - 1):
        ti, bi = draws[r, :n], draws[r, n:]
        Xt, Yn = X[ti], obs["Ybscan"][bi]
        M = orc.create_multiblock(Xt, co, "mb", bscan, 0, Xbscan=obs["Xbscan"], Ybscan=Yn)          # :392
        s_hat = np.sqrt(np.sum((M.T @ U) ** 2, axis=0))                                               # :404-405
        raw = orc.create_multiblock(Xt, co, "mb", bscan, 0, norm_opt=False, Xbscan=obs["Xbscan"], Ybscan=Yn)
        want = np.sqrt(s_hat ** 4 / np.sum(s_hat ** 4) * np.sum(raw ** 2))                           # :413-424 (Q3)
        assert_close(s_list[r][live], want[live], 1e-9, 0, f"mb perm s_hat[{r}]")
        # the two latent variables that are null in the OBSERVED block are not null in a permuted one (the null
        # vectors of the row-normalised block are D w, and D changes with the sample): 1e-6 of the largest here, a
        # quadratic form of the Gram resolves them to eps k s_1^2 / s_j^2
        assert (want[~live] > 1e-8 * want[0]).all()
        assert_close(s_list[r][~live], want[~live], 1e-6, 0, f"mb perm s_hat[{r}], observed-null variables")
    org_s = rt.perm_debug_dict["org_s"]
    np.testing.assert_array_equal(rt.permute_ratio, (s_list >= org_s).sum(0) / (nperm + 1))


def test_config6_mb_bootstrap_full_size():
    """Multiblock bootstrap (not a BASELINE config; SURVEY a12 on config 4's data): rows (K2s ROWS variant),
    projection (K4m) + K5x at n = 120, kr = 38."""
    from plspy_amd.bootstrap_permutation import ResampleTest
    co = np.array([[20] * 3, [20] * 3])
    X = np.random.RandomState(0).randn(120, 200_000)
    Y = np.random.RandomState(1).randn(120, 8)
    bscan = [1, 2]
    obs = orc.observed("mb", X, co, Y=Y, mctype=0, bscan=bscan)
    U, s, V = obs["U"], obs["s"], obs["V"]
    nboot = 14
    np.random.seed(5)
    from plspy_amd.engine import ProjectionEngine
    eng = ProjectionEngine(X)
    rt = ResampleTest._create("mb", X, Y, U, s.copy(), V, co, 0, nperm=0, nboot=nboot, bscan=bscan,
                              Xbscan=obs["Xbscan"], Ybscan=obs["Ybscan"], lvcorrs_orig=obs["lvcorrs"],
                              Tvsc_orig=obs["Tvsc_orig"], engine=eng)
    # round 3: the un-normalised rows from the two-stage kernel (plsr_split_rows), the projection as a stream
    # over them (plsr_rows_project)
    assert eng.last_item_kernel == "rows+project", eng.last_item_kernel
    assert eng.last_latent_kernel == "index"          # (K5i: the behaviour sample's rows + the raw task rows)
    draws = rt.boot_debug_dict["indices"]
    n = 120
    ti, bi = draws[:, :n], draws[:, n:]
    live = s > 1e-10 * s.max()
    sub = _subset(200_000, 200)
    VSsub = []
    for b in range(nboot):
        M = orc.create_multiblock(X[ti[b]], co, "mb", bscan, 0, Xbscan=obs["Xbscan"][bi[b]],
                                  Ybscan=obs["Ybscan"][bi[b]])       # :610
        VSsub.append(M[:, sub].T @ U)                                 # :620 on a voxel subset (rows normalised over ALL voxels)
        if b not in (0, nboot - 1):
            continue
        Vh = orc.normalize(M.T @ U)                                   # :620, :623
        lc = orc.compute_corr(obs["Xbscan"][bi[b]] @ Vh, obs["Ybscan"][bi[b]], co[:, bscan])
        assert_close(rt.LVcorr[b][:, live], lc[:, live], 1e-8, 1e-11, f"mb LVcorr[{b}]")
        Td = orc.group_condition_means(orc.calculate_smeanmat(X[ti[b]], co, 0) @ Vh, co)
        assert_close(rt.boot_debug_dict["Tdistrib"][b][:, live], Td[:, live], 1e-8, 1e-11, f"mb Tdistrib[{b}]")
    # :695, :700: np.std of every bootstrap's projection, on the subset
    want_sd = np.std(np.array(VSsub), axis=0)
    assert_close(rt.std_errs[sub][:, live], want_sd[:, live], 1e-8, 1e-12, "mb std_errs")
    assert_close(rt.boot_ratios[sub][:, live], ((V * s)[sub] / want_sd)[:, live], 1e-7, 1e-9, "mb boot_ratios")
    assert np.isfinite(rt.std_errs).all()


# ---------------------------------------------------------------------------
# voxels with a strong, stable effect: |boot ratio| = |V s| / std_errs of 30 and of 1000
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("groups", [(10, 10), (12, 12)])        # n = 60: K1br (X in registers); n = 72: LDS-fed kernel
def test_std_errs_of_high_ratio_voxels(groups):
    """The bootstrap kernels accumulate plain sums of VS and VS^2 and shift them by the observed VS at
    the merge (sum (x - ref)^2 = sum x^2 - 2 ref sum x + R ref^2), which cancels (ref / sd)^2 of the 2^53.
    Planted: two blocks of voxels whose group x condition effect is 30 and 1000 times the noise -- the
    voxels a user looks at.  std_errs against np.std of the directly projected bootstraps
    (bootstrap_permutation.py:695-703)."""
    import torch
    from plspy_amd import operators, resample
    from plspy_amd.engine import ProjectionEngine
    nc, p, R = 3, 20_000, 1000
    co = np.array([[g] * nc for g in groups])
    n = int(co.sum())
    rs = np.random.RandomState(17)
    X = rs.randn(n, p)
    cell = np.repeat(np.arange(co.size), co.reshape(-1))
    effect = rs.randn(co.size)[cell]
    blocks = {30.0: np.arange(1000, 1200), 1000.0: np.arange(9000, 9200)}
    for amp, cols in blocks.items():
        X[:, cols] += 0.12 * amp * effect[:, None] * (0.5 + rs.rand(len(cols)))[None, :]
    W = operators.mean_centre_operator(co, 0)
    Wm = operators.cell_mean_operator(co)
    U, s, Vt = np.linalg.svd(W @ X, full_matrices=False)
    s[np.abs(s) < 1e-12] = 0
    V = Vt.T
    k = U.shape[1]
    live = s > 1e-10 * s.max()
    eng = ProjectionEngine(X)
    assert eng.plan(k, R, k2=k, boot=True)["register_resident"] == (n <= 64)
    M = W.T @ U
    ref = V * s
    np.random.seed(99)
    binds = resample.bootstraps(co, R)
    res = eng.boot_phase(k, inds=eng.dev(binds, torch.int32), M=eng.dev(M), ref=eng.dev(ref), Xm=eng.apply_operator(Wm))
    sd, ratio = eng.boot_finalize(res["S1"], res["S2"], R, num=eng.dev(ref))
    sd, ratio = sd.cpu().numpy(), ratio.cpu().numpy()
    a = _operators(M, binds)
    for amp, cols in blocks.items():
        VS = np.einsum("iv,bij->bvj", X[:, cols], a)
        want = np.std(VS, axis=0)
        r = np.abs(ref[cols] / want)[:, live]
        assert amp < r.max() < 3 * amp, (amp, r.max())              # the block really holds ratios of that size
        assert_close(sd[cols][:, live], want[:, live], 1e-9, 0, f"std_errs, |ratio| up to {r.max():.0f}")
        assert_close(ratio[cols][:, live], (ref[cols] / want)[:, live], 1e-9, 0, f"boot_ratios, |ratio| up to {r.max():.0f}")
