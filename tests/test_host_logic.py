"""CPU-side pieces of the product: index draws, operators, the C-ABI's host
entry points, sharding arithmetic.  No GPU needed."""
import ctypes
import os
import re

import numpy as np
import pytest

from oracle import plspy_oracle as orc
from plspy_amd import _build, _lib, dist, operators, resample
from tests._util import golden_names, load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    _build.build()
    return _lib.load()


def test_abi_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "plsr.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(plsr_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.plsr_abi_version() == 2
    assert lib.plsr_strerror(0) == b"ok"


@pytest.mark.parametrize("k,kp,period", [(1, 1, 1), (2, 2, 1), (3, 3, 3), (4, 4, 1), (5, 5, 5),
                                         (6, 6, 3), (7, 8, 2), (9, 12, 3), (12, 12, 3), (24, 24, 6)])
def test_layout(lib, k, kp, period):
    """Quad layout (n > 64: the LDS-fed kernels)."""
    lay = _lib.Layout()
    assert lib.plsr_layout_init(100, k, 10, ctypes.byref(lay)) == 0
    assert (lay.kp, lay.period, lay.nk, lay.Rp) == (kp, period, 25, 12)
    assert (4 * lay.period) % lay.kp == 0          # the slot -> latent-variable map repeats
    assert lay.ntiles == -(-(lay.kp * lay.Rp // 4) // 4)
    assert lay.frag_elems == lay.ntiles * lay.nk * 64 + 4 * 64      # + prefetch padding


@pytest.mark.parametrize("n,k,R", [(60, 6, 10), (13, 1, 16), (64, 7, 1000), (16, 24, 17)])
def test_layout_lv_major(lib, n, k, R):
    """n <= 64 (4..16 k-steps): LV-major layout of the register-resident kernels,
    marked by period 0 -- tile t = 16 consecutive resamples of latent variable t / (Rp/16)."""
    lay = _lib.Layout()
    assert lib.plsr_layout_init(n, k, R, ctypes.byref(lay)) == 0
    assert (lay.kp, lay.period, lay.nk) == (k, 0, -(-n // 4))
    assert lay.Rp == -(-R // 16) * 16 and lay.ntiles == k * lay.Rp // 16
    assert lay.frag_elems == lay.ntiles * lay.nk * 64 + 4 * 64


def test_layout_rejects(lib):
    lay = _lib.Layout()
    assert lib.plsr_layout_init(0, 6, 10, ctypes.byref(lay)) == -1
    assert lib.plsr_layout_init(60, 6, 0, ctypes.byref(lay)) == -1
    assert lib.plsr_layout_init(2000, 6, 10, ctypes.byref(lay)) == -2     # X tile > LDS
    assert lib.plsr_batch_workspace_bytes(None, 10, 0) == 0


@pytest.mark.parametrize("mctype", [0, 1, 2, 3])
@pytest.mark.parametrize("groups,nc", [((10, 10), 3), ((3, 2), 2), ((8,), 3), ((4, 5, 6), 2)])
def test_mean_centre_operator_matches_oracle(groups, nc, mctype):
    co = np.array([[g] * nc for g in groups])
    n = co.sum()
    X = np.random.RandomState(1).randn(n, 17)
    W = operators.mean_centre_operator(co, mctype)
    Wm = operators.cell_mean_operator(co)
    means, mc = orc.mean_centre(X, co, mctype)
    np.testing.assert_allclose(W @ X, mc, rtol=0, atol=2e-15)
    np.testing.assert_allclose(Wm @ X, means, rtol=0, atol=2e-15)
    # resample folds into the operator (appendix A2)
    inds = np.random.RandomState(2).randint(0, n, n)
    P = np.zeros((n, n))
    P[np.arange(n), inds] = 1
    np.testing.assert_allclose((W @ P) @ X, orc.mean_centre(X[inds], co, mctype)[1], atol=3e-15)


@pytest.mark.parametrize("name", golden_names("mct_g"))
def test_index_draws_match_reference(name):
    """Same seed -> the same resamples as the reference drew (the fixture holds
    the reference's raw np.random outputs)."""
    fx = load_golden(name)
    co = fx["cond_order"]
    np.random.seed(fx["seed"])
    perm = resample.task_permutations(co, fx["nperm"])
    boot = resample.bootstraps(co, fx["nboot"])
    np.random.seed(fx["seed"])
    smp = orc.Sampler()
    for i in range(fx["nperm"]):
        np.testing.assert_array_equal(perm[i], smp.perm_task(co))
    for i in range(fx["nboot"]):
        np.testing.assert_array_equal(boot[i], smp.boot(co))
    # and against the raw reference stream: the last per-condition shuffle of
    # the first permutation is a permutation of that condition's rows
    nsub, nc = sum(fx["groups"]), fx["ncond"]
    raw = fx["draws"]
    first = np.stack(raw[nsub:nsub + nc]).ravel()
    np.testing.assert_array_equal(perm[0], first)
    # bootstrap rows: per group, reference's choice() output indexes the table
    off = fx["nperm"] * (nsub + nc)
    tables = resample.subject_tables(co)
    expect = np.concatenate([t[raw[off + g]].T.ravel() for g, t in enumerate(tables)])
    np.testing.assert_array_equal(boot[0], expect)


def test_shard_bounds_cover():
    for R in (0, 1, 7, 1000):
        for n in (1, 2, 3, 8):
            b = [dist.shard_bounds(R, r, n) for r in range(n)]
            assert b[0][0] == 0 and b[-1][1] == R
            assert all(b[i][1] == b[i + 1][0] for i in range(n - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def test_assert_close_rejects_non_finite_mismatch():
    """The parity helper must not let a NaN / inf from a kernel pass against a finite
    expected value (and must still accept matching non-finite entries)."""
    import pytest
    from tests._util import assert_close
    assert_close([1.0, 2.0], [1.0, 2.0 + 1e-13], 1e-10)
    assert_close([np.nan, 1.0, np.inf, -np.inf], [np.nan, 1.0, np.inf, -np.inf], 1e-10)
    for got, want in (([np.nan, 1.0], [2.0, 1.0]), ([2.0, 1.0], [np.nan, 1.0]), ([np.inf, 1.0], [2.0, 1.0]),
                      ([np.inf], [-np.inf]), ([1.0], [1.1])):
        with pytest.raises(AssertionError):
            assert_close(got, want, 1e-10)


def test_degenerate_guard_matches_full_test():
    """cf.degenerate_guard (index shortcut) == any_group_std_zero on the gathered stack: distinct
    columns (bootstrap draws that repeat one row through a group), columns with repeated values,
    and the tolerant slicing of a cond_order longer than the rows (quirk Q8)."""
    from plspy_amd import class_functions as cf
    rs = np.random.RandomState(5)
    co = np.array([[2, 2], [3, 1]])
    n = int(co.sum())
    for Y in (rs.randn(n, 3), np.round(rs.randn(n, 3)), np.where(rs.rand(n, 3) < 0.2, np.nan, rs.randn(n, 3))):
        rows = rs.randint(0, n, size=(400, n))
        rows[5, :4] = 3                      # first group: one row repeated
        rows[9, 4:] = 0                      # second group
        rows[11] = np.arange(n)
        want = cf.any_group_std_zero(Y[rows], co)
        assert np.array_equal(cf.degenerate_guard(Y, co)(rows), want)
        assert want[5] and want[9] or np.isnan(Y).any()
    # bscan subset: fewer rows than cond_order describes
    Yb = rs.randn(5, 2)
    rows = rs.randint(0, 5, size=(300, 5))
    rows[7, :4] = 2
    assert np.array_equal(cf.degenerate_guard(Yb, co)(rows), cf.any_group_std_zero(Yb[rows], co))


def test_workspace_queries_refuse_what_the_calls_refuse(lib):
    """The kernels that address rows by 32-bit byte offsets: their workspace queries must return 0 exactly
    for the shapes the calls would refuse, so that the engine's fallbacks are taken before anything is
    launched (rows p apart, as the engine passes them).  plsr_latent walks an X of 4 GiB or more in blocks
    of rows instead of refusing it."""
    # K5: X = 240 x 2.3 M doubles = 4.4 GB -> served (row blocks); VS^T of one group beyond 4 GiB -> refused
    assert lib.plsr_latent_workspace_bytes(240, 12, 10, 2_300_000) > 0
    assert lib.plsr_latent_workspace_bytes(120, 48, 4, 4_600_000) > 0
    assert lib.plsr_latent_workspace_bytes(240, 12, 10, 50_000_000) == 0          # 12 x 50 M x 8 B >= 4 GiB
    assert lib.plsr_latent_workspace_bytes(240, 12, 10, 40_000_000) == 0          # fewer than 16 rows per block
    # K4a / K4b: the result stores' lane offsets span 13 rows of VS^T
    cells = (ctypes.c_int32 * 3)(0, 20, 40)
    z = (ctypes.c_int32 * 2)(1, 1)
    lo = (ctypes.c_int32 * 2)(0, 20)
    hi = (ctypes.c_int32 * 2)(20, 40)
    assert lib.plsr_item_agg_workspace_bytes(40, 40, 16, cells, z, lo, hi, 2, 8, 200_000, 1, 0) > 0
    assert lib.plsr_item_agg_workspace_bytes(40, 40, 16, cells, z, lo, hi, 2, 8, 42_000_000, 1, 0) == 0
    assert lib.plsr_item_beh_workspace_bytes(40, 40, 8, 16, cells, lo, hi, 2, 8, 200_000, 1) > 0
    assert lib.plsr_item_beh_workspace_bytes(40, 40, 8, 16, cells, lo, hi, 2, 8, 42_000_000, 1) == 0
    # K5i: 32-bit offsets into an item's VS^T; at most 128 rows of X; the caller's bound on different rows <= n
    assert lib.plsr_latent_index_workspace_bytes(120, 48, 125, 200_000, 120, 96, 0) > 0
    assert lib.plsr_latent_index_workspace_bytes(120, 38, 125, 200_000, 80, 64, 6) > 0
    assert lib.plsr_latent_index_workspace_bytes(120, 48, 4, 11_200_000, 120, 96, 0) == 0
    assert lib.plsr_latent_index_workspace_bytes(129, 48, 4, 200_000, 120, 96, 0) == 0
    assert lib.plsr_latent_index_workspace_bytes(120, 48, 4, 200_000, 120, 121, 0) == 0
    assert lib.plsr_latent_xb_bytes(120, 200_000) == 200_000 * 120 * 8 and lib.plsr_latent_xb_bytes(129, 200_000) == 0
    # K2s: 32-bit row offsets into X
    rows = (ctypes.c_int32 * 4)(10, 10, 10, 10)
    assert lib.plsr_split_gram_workspace_bytes(40, 200_000, 200_000, 8, rows, 4, 4, 0, 64, 5) > 0
    assert lib.plsr_split_gram_workspace_bytes(240, 2_300_000, 2_300_000, 8, rows, 4, 4, 0, 64, 5) == 0
    assert lib.plsr_split_gram_workspace_bytes(40, 200_000, 200_000, 9, rows, 4, 4, 0, 72, 5) == 0    # b > 8
    assert lib.plsr_split_gram_workspace_bytes(40, 8, 8, 8, rows, 4, 4, 0, 64, 5) == 0               # p < 16
