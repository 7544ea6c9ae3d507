"""GPU parity for behaviour (rb) and multiblock (mb) PLS: the gather + z-score
kernel (K3), observed decomposition, permutation test and split-half tests
against the golden vectors from the reference and the oracle."""
import numpy as np
import pytest

from oracle import plspy_oracle as orc
from tests._util import assert_close, load_golden, nonnull

pytestmark = pytest.mark.gpu


def _align(mine, ref, live):
    return np.sign(np.sum(mine[:, live] * ref[:, live], axis=0))


@pytest.mark.parametrize("n,p,cells", [(22, 130, [0, 6, 11, 17, 22]), (9, 1, [0, 2, 9]), (40, 257, [0, 40])])
def test_gather_zscore_matches_oracle(n, p, cells):
    from plspy_amd.engine import ProjectionEngine
    rs = np.random.RandomState(n)
    X = rs.randn(n, p) * 3 + 50.0          # large mean / small spread: cancellation check
    X[:, p // 2] = 7.0                      # a constant voxel -> 0 after z-scoring
    eng = ProjectionEngine(X)
    src = np.stack([rs.randint(0, n, n), np.arange(n), rs.permutation(n)])
    cell_z = np.ones(len(cells) - 1, dtype=int)
    cell_z[0] = 0 if len(cells) > 2 else 1
    out = eng.gather_zscore(src, cells, cell_z).cpu().numpy()
    for it in range(3):
        G = X[src[it]]
        for c in range(len(cells) - 1):
            lo, hi = cells[c], cells[c + 1]
            if cell_z[c]:
                ref = orc._zscore_cell(G[lo:hi]) / np.sqrt(hi - lo)
                ref = np.nan_to_num(ref)
                # duplicated rows can make a voxel constant inside a cell
                np.testing.assert_allclose(out[it, lo:hi], ref, rtol=1e-10, atol=1e-12)
            else:
                np.testing.assert_array_equal(out[it, lo:hi], G[lo:hi])


@pytest.mark.parametrize("name", ["rb_g6x5_c2_b3", "rb_split_g6x6_c2_b2"])
def test_rb_observed_and_perm(name):
    import plspy_amd
    fx = load_golden(name)
    np.random.seed(fx["seed"])
    res = plspy_amd.PLS(fx["X"].copy(), fx["groups"], fx["ncond"], Y=fx["Y"].copy(),
                        num_perm=fx["nperm"], num_boot=0, pls_method="rb")
    live = nonnull(fx)
    assert_close(res.R, fx["R"], 1e-10, 1e-12, "R")
    assert_close(res.s[live], fx["s"][live], 1e-10, 0, "s")
    sign = _align(res.V, fx["U"], live)           # after the swap res.V is the k x k U
    assert_close(res.V[:, live] * sign, fx["U"][:, live], 1e-7, 1e-9, "U")
    assert_close(res.U[:, live] * sign, fx["V"][:, live], 1e-7, 1e-9, "V")
    assert_close(res.lvcorrs[:, live] * sign, fx["lvcorrs"][:, live], 1e-7, 1e-9, "lvcorrs")
    assert_close(res.X_latent[:, live] * sign, fx["X_latent"][:, live], 1e-7, 1e-9, "X_latent")
    rt = res.resample_tests
    assert_close(rt.perm_debug_dict["s_list"][:, live], fx["s_list"][:, live], 1e-9, 1e-11, "s_list")
    n1 = fx["nperm"] + 1
    np.testing.assert_array_equal(np.rint(rt.permute_ratio * n1)[live], np.rint(fx["permute_ratio"] * n1)[live])
    np.testing.assert_array_equal(np.rint(rt.stepdown_ratio * n1)[live], np.rint(fx["stepdown_ratio"] * n1)[live])


def test_rb_perm_seam_with_reference_svd():
    """The resample seam fed the reference's own U/s/V: all k latent variables
    comparable, counts exact."""
    from plspy_amd.bootstrap_permutation import ResampleTest
    fx = load_golden("rb_g6x5_c2_b3")
    np.random.seed(fx["seed"])
    rt = ResampleTest._create("rb", fx["X"], fx["Y"], fx["U"], fx["s"].copy(), fx["V"], fx["cond_order"],
                              None, nperm=fx["nperm"], nboot=0)
    assert_close(rt.perm_debug_dict["s_list"], fx["s_list"], 1e-10, 1e-11, "s_list")
    n1 = fx["nperm"] + 1
    np.testing.assert_array_equal(np.rint(rt.permute_ratio * n1), np.rint(fx["permute_ratio"] * n1))
    np.testing.assert_array_equal(np.rint(rt.stepdown_ratio * n1), np.rint(fx["stepdown_ratio"] * n1))


@pytest.mark.parametrize("name", ["mb_g6x6_c3_b2", "mb_split_g6x5_c3_b2"])
def test_mb_observed_and_perm(name):
    import plspy_amd
    fx = load_golden(name)
    np.random.seed(fx["seed"])
    res = plspy_amd.PLS(fx["X"].copy(), fx["groups"], fx["ncond"], Y=fx["Y"].copy(), mctype=fx["mctype"],
                        num_perm=fx["nperm"], num_boot=0, pls_method="mb", bscan=fx["bscan"])
    live = nonnull(fx)
    assert_close(res.multiblock, fx["multiblock"], 1e-10, 1e-12, "multiblock")
    assert_close(res.s[live], fx["s"][live], 1e-10, 0, "s")
    sign = _align(res.V, fx["U"], live)
    assert_close(res.U[:, live] * sign, fx["V"][:, live], 1e-7, 1e-9, "V")
    assert_close(res.Tusc[:, live] * sign, fx["Tusc"][:, live], 1e-7, 1e-9, "Tusc")
    assert_close(res.Busc[:, live] * sign, fx["Busc"][:, live], 1e-7, 1e-9, "Busc")
    assert_close(res.lvcorrs[:, live] * sign, fx["lvcorrs"][:, live], 1e-7, 1e-9, "lvcorrs")
    rt = res.resample_tests
    n1 = fx["nperm"] + 1
    np.testing.assert_array_equal(np.rint(rt.permute_ratio * n1)[live], np.rint(fx["permute_ratio"] * n1)[live])
    np.testing.assert_array_equal(np.rint(rt.stepdown_ratio * n1)[live], np.rint(fx["stepdown_ratio"] * n1)[live])


def test_mb_perm_against_oracle_values():
    """s_hat of every multiblock permutation against the oracle on the same
    draws (the reference does not export them for mb)."""
    from plspy_amd.bootstrap_permutation import ResampleTest
    fx = load_golden("mb_g6x6_c3_b2")
    co = fx["cond_order"]
    obs = orc.observed("mb", fx["X"], co, Y=fx["Y"], mctype=fx["mctype"], bscan=fx["bscan"])
    np.random.seed(fx["seed"])
    ref = orc.permutation_test("mb", fx["X"], fx["Y"], fx["U"], fx["s"].copy(), fx["V"], co, fx["mctype"],
                               fx["nperm"], bscan=fx["bscan"], Xbscan=obs["Xbscan"], Ybscan=obs["Ybscan"])
    np.random.seed(fx["seed"])
    rt = ResampleTest._create("mb", fx["X"], fx["Y"], fx["U"], fx["s"].copy(), fx["V"], co, fx["mctype"],
                              nperm=fx["nperm"], nboot=0, bscan=fx["bscan"], Xbscan=obs["Xbscan"],
                              Ybscan=obs["Ybscan"])
    assert_close(rt.perm_debug_dict["s_list"], ref["s_list"], 1e-9, 1e-11, "mb s_list")
    np.testing.assert_array_equal(rt.permute_ratio, ref["permute_ratio"])
    np.testing.assert_array_equal(rt.stepdown_ratio, ref["stepdown_ratio"])


def _seam(fx, **kw):
    from plspy_amd.bootstrap_permutation import ResampleTest
    co = fx["cond_order"]
    alg = fx["method"]
    obs = orc.observed(alg, fx["X"], co, Y=fx["Y"], mctype=fx["mctype"], bscan=fx["bscan"])
    if alg == "rb":
        extra = dict(lvcorrs_orig=orc.compute_corr(fx["X"] @ fx["V"], fx["Y"], co))
    else:
        extra = dict(bscan=fx["bscan"], Xbscan=obs["Xbscan"], Ybscan=obs["Ybscan"],
                     lvcorrs_orig=orc.compute_corr(obs["Xbscan"] @ fx["V"], obs["Ybscan"], co[:, fx["bscan"]]),
                     Tvsc_orig=orc.group_condition_means(fx["X"] @ orc.normalize(fx["V"]), co))
    np.random.seed(fx["seed"])
    return ResampleTest._create(alg, fx["X"], fx["Y"], fx["U"], fx["s"].copy(), fx["V"], co, fx["mctype"],
                                nperm=fx["nperm"], nboot=fx["nboot"], **extra, **kw)


@pytest.mark.parametrize("name", ["rb_g6x5_c2_b3", "rb_split_g6x6_c2_b2", "mb_g6x6_c3_b2", "mb_split_g6x5_c3_b2"])
def test_bootstrap_rb_mb_against_reference(name):
    """Resample seam with the reference's own U/s/V: perm then boot consume the
    RNG exactly like the reference, every bootstrap statistic is compared."""
    fx = load_golden(name)
    rt = _seam(fx)
    live = nonnull(fx)
    assert_close(rt.std_errs[:, live], fx["std_errs"][:, live], 1e-9, 1e-13, "std_errs")
    assert_close(rt.boot_ratios[:, live], fx["boot_ratios"][:, live], 1e-8, 1e-10, "boot_ratios")
    assert_close(rt.LVcorr[:, :, live], fx["LVcorr"][:, :, live], 1e-8, 1e-11, "LVcorr")
    assert_close(rt.conf_ints[0][:, live], fx["conf_lo"][:, live], 1e-8, 1e-11, "conf lo")
    assert_close(rt.conf_ints[1][:, live], fx["conf_hi"][:, live], 1e-8, 1e-11, "conf hi")
    if fx["method"] == "mb":
        assert_close(rt.conf_ints_T[0][:, live], fx["confT_lo"][:, live], 1e-8, 1e-11, "confT lo")
        assert_close(rt.conf_ints_T[1][:, live], fx["confT_hi"][:, live], 1e-8, 1e-11, "confT hi")
    n1 = fx["nperm"] + 1
    np.testing.assert_array_equal(np.rint(rt.permute_ratio * n1), np.rint(fx["permute_ratio"] * n1))


@pytest.mark.parametrize("name", ["rb_g6x5_c2_b3", "mb_g6x6_c3_b2"])
def test_full_pls_rb_mb_with_bootstrap(name):
    import plspy_amd
    fx = load_golden(name)
    np.random.seed(fx["seed"])
    kw = dict(Y=fx["Y"].copy(), num_perm=fx["nperm"], num_boot=fx["nboot"], pls_method=fx["method"])
    if fx["method"] == "mb":
        kw.update(mctype=fx["mctype"], bscan=fx["bscan"])
    res = plspy_amd.PLS(fx["X"].copy(), fx["groups"], fx["ncond"], **kw)
    live = nonnull(fx)
    sign = _align(res.V, fx["U"], live)
    rt = res.resample_tests
    assert_close(rt.std_errs[:, live], fx["std_errs"][:, live], 1e-7, 1e-11, "std_errs")
    assert_close(rt.boot_ratios[:, live] * sign, fx["boot_ratios"][:, live], 1e-6, 1e-9, "boot_ratios")
    assert_close(rt.conf_ints[0][:, live] * sign, np.where(sign > 0, fx["conf_lo"][:, live], fx["conf_hi"][:, live]),
                 1e-6, 1e-9, "conf ints (sign aligned)")


def _replay_rng_until_split(fx):
    """Consume np.random exactly as the reference's perm + boot loops do."""
    co = fx["cond_order"]
    alg = fx["method"]
    np.random.seed(fx["seed"])
    obs = orc.observed(alg, fx["X"], co, Y=fx["Y"], mctype=fx["mctype"], bscan=fx["bscan"])
    kw = dict(bscan=fx["bscan"], Xbscan=obs.get("Xbscan"), Ybscan=obs.get("Ybscan"))
    import warnings
    with warnings.catch_warnings(), np.errstate(all="ignore"):
        warnings.simplefilter("ignore")
        perm = orc.permutation_test(alg, fx["X"], fx["Y"], fx["U"], fx["s"].copy(), fx["V"], co,
                                    fx["mctype"], fx["nperm"], **kw)
        orc.bootstrap_test(alg, fx["X"], fx["Y"], fx["U"], perm["s"], fx["V"], co, fx["mctype"], fx["nboot"],
                           lvcorrs_orig=fx.get("lvcorrs"), Tvsc_orig=np.zeros((co.size, len(fx["s"]))), **kw)
    return obs


@pytest.mark.parametrize("name", ["rb_split_g6x6_c2_b2", "mb_split_g6x5_c3_b2"])
def test_split_half_rb_mb(name):
    from plspy_amd import split_half_resampling as sh
    fx = load_golden(name)
    obs = _replay_rng_until_split(fx)
    alg, co, S, lv = fx["method"], fx["cond_order"], fx["num_split"], fx["lv"]
    kw = dict(mctype=fx["mctype"], bscan=fx["bscan"], Xbscan=obs.get("Xbscan"), Ybscan=obs.get("Ybscan"))
    tt = sh.split_half_test_train(alg, fx["X"], fx["Y"], co, S, **kw)
    res = sh.split_half(alg, fx["X"], fx["Y"], co, S, lv=lv, CI=0.95, **kw)
    # compare the leading latent variables whose singular values are well
    # separated from the null space in every split
    ref_train = fx["tt_pls_s_train"]
    nl = int(np.sum(ref_train[0, :, :].min(axis=1) > 1e-8 * ref_train[0, 0, :].max()))
    nl = max(lv, min(nl, 4))
    for key in ("pls_s_train", "pls_s_train_null"):
        assert_close(tt[key][:, :nl, :], fx["tt_" + key][:, :nl, :], 1e-8, 1e-11, key)
    d = np.arange(nl)
    for key in ("pls_s_test", "pls_s_test_null"):
        assert_close(tt[key][d, d, :], fx["tt_" + key][d, d, :], 1e-6, 1e-9, key + " diag")
        assert_close(np.abs(tt[key][:nl, :nl, :]), np.abs(fx["tt_" + key][:nl, :nl, :]), 1e-6, 1e-9, key)
    for key in ("pls_dist_u", "pls_dist_v", "pls_dist_null_u", "pls_dist_null_v"):
        assert_close(np.abs(res[key][:nl, :nl, :]), np.abs(fx["sh_" + key][:nl, :nl, :]), 1e-6, 1e-9, key)
    for key, val in res.items():
        if not key.startswith("pls_dist"):
            assert_close(np.array(val)[:lv], fx["sh_" + key][:lv], 1e-5, 1e-9, key)


@pytest.mark.parametrize("shape", [
    # (n, p, cells (row counts), z flags, k, items)
    (24, 130, (6, 5, 7, 6), (1, 1, 1, 1), 5, 7),          # ragged cells, k < 16, voxels off the tile
    (40, 257, (40, 12, 9, 11), (0, 1, 1, 1), 20, 5),      # copy cell + z cells (multiblock layout), 2 LV tiles
    (30, 64, (10, 10, 10), (1, 1, 1), 48, 9),             # 3 LV tiles
    (21, 70, (21,), (1,), 70, 3),                         # one cell, 5 LV tiles
])
def test_fused_items_match_gather_then_project(shape):
    """K4f (gather + z-score + projection in one pass over an LDS-resident X
    tile) against the NumPy statement of the same thing and against the
    two-kernel path (plsr_gather_zscore + dense product)."""
    import torch
    from plspy_amd.engine import ProjectionEngine
    n, p, cells, zflags, k, items = shape
    rs = np.random.RandomState(n + p)
    X = rs.randn(n, p) * 3 + rs.randn(1, p) * 50 + 100
    X[:, 3] = 7.0                                     # a constant voxel -> z cells give 0
    nz = sum(cells)
    cell_lo = np.concatenate(([0], np.cumsum(cells)))
    src = rs.randint(0, n, size=(items, nz)).astype(np.int32)
    rows = rs.randn(items, k, nz)
    ref = rs.randn(p, k)
    eng = ProjectionEngine(X)
    S1 = torch.zeros((p, k), dtype=torch.float64, device=eng.device)
    S2 = torch.zeros_like(S1)
    vst, rowsq = eng.item_fused(src, cell_lo, zflags, rows, ref=ref, S1=S1, S2=S2, want_vst=True,
                                want_rowsq=True)
    Z = eng.gather_zscore(src, cell_lo, zflags).cpu().numpy()       # (items, nz, p), K3
    # NumPy statement
    Zn = np.empty((items, nz, p))
    for b in range(items):
        G = X[src[b]]
        for c, (lo, hi) in enumerate(zip(cell_lo[:-1], cell_lo[1:])):
            if zflags[c]:
                blk = G[lo:hi]
                mu, sd = blk.mean(0), blk.std(0)
                with np.errstate(divide="ignore", invalid="ignore"):
                    z = (blk - mu) / sd / np.sqrt(hi - lo)
                z[:, sd <= 2.220446049250313e-16 * np.abs(mu)] = 0.0
                Zn[b, lo:hi] = z
            else:
                Zn[b, lo:hi] = G[lo:hi]
    np.testing.assert_allclose(Z, Zn, rtol=1e-9, atol=1e-12)
    want = np.einsum("bji,biv->bjv", rows, Zn)
    got = vst.cpu().numpy()
    scale = np.abs(want).max()
    np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-11 * scale)
    np.testing.assert_allclose(rowsq.cpu().numpy(), (want ** 2).sum(-1), rtol=1e-10)
    d = np.transpose(want, (0, 2, 1)) - ref
    np.testing.assert_allclose(S1.cpu().numpy(), d.sum(0), rtol=1e-9, atol=1e-10 * scale)
    np.testing.assert_allclose(S2.cpu().numpy(), (d ** 2).sum(0), rtol=1e-9, atol=1e-10 * scale ** 2)
