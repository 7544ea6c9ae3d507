"""GPU parity for K2 (Gram + Jacobi thin SVD) and the split-half tests."""
import numpy as np
import pytest

from tests._util import assert_close, load_golden, run_oracle_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng_factory():
    from plspy_amd.engine import ProjectionEngine
    return ProjectionEngine


@pytest.mark.parametrize("n,p,m,S", [(11, 70, 6, 5), (60, 1000, 12, 9), (60, 333, 16, 3), (45, 129, 20, 2),
                                     (120, 500, 76, 2), (120, 64, 96, 1), (240, 300, 24, 3), (7, 1, 3, 2)])
def test_gram_matches_numpy(eng_factory, n, p, m, S):
    rs = np.random.RandomState(n + m)
    X = rs.randn(n, p) + 0.3
    rows = rs.randn(S, m, n)
    eng = eng_factory(X)
    G = eng.gram_phase(rows).cpu().numpy()
    mm = (m + 15) // 16 * 16
    assert G.shape == (S, mm, mm)
    for s in range(S):
        M = rows[s] @ X
        ref = M @ M.T
        scale = np.abs(ref).max()
        np.testing.assert_allclose(G[s, :m, :m], ref, rtol=0, atol=5e-13 * scale)
        assert np.all(G[s, m:, :] == 0) and np.all(G[s, :, m:] == 0)


@pytest.mark.parametrize("k", [1, 2, 5, 6, 12, 37, 48, 64])
def test_eigh_matches_numpy(eng_factory, k):
    import torch
    rs = np.random.RandomState(k)
    eng = eng_factory(rs.randn(4, 8))
    S, off, mm = 5, 3, 80
    G = np.zeros((S, mm, mm))
    for s in range(S):
        A = rs.randn(k, k + 3)
        if s == 1 and k > 2:
            A[-1] = A[0]                      # exactly rank-deficient
        if s == 2:
            A *= np.logspace(0, -5, k)[:, None]   # graded spectrum
        G[s, off:off + k, off:off + k] = A @ A.T
    ev, vec = eng.eigh(torch.as_tensor(G, device=eng.device), off, k)
    ev, vec = ev.cpu().numpy(), vec.cpu().numpy()
    for s in range(S):
        B = G[s, off:off + k, off:off + k]
        w = np.linalg.eigvalsh(B)[::-1]
        np.testing.assert_allclose(ev[s], w, rtol=1e-12, atol=1e-14 * w[0])
        V = vec[s]
        np.testing.assert_allclose(V.T @ V, np.eye(k), atol=1e-13)
        np.testing.assert_allclose(V @ np.diag(ev[s]) @ V.T, B, atol=1e-13 * w[0])


@pytest.mark.parametrize("name", ["mct_g10x10_c3_mc0", "mct_g10x10_c3_mc2", "mct_g3x2_c2"])
def test_thin_svd_against_lapack(eng_factory, name):
    from plspy_amd import operators
    fx = load_golden(name)
    eng = eng_factory(fx["X"])
    W = operators.mean_centre_operator(fx["cond_order"], fx["mctype"])
    U, s, V = eng.thin_svd(W)
    live = fx["s"] > 1e-10 * fx["s"].max()
    np.testing.assert_allclose(s[live], fx["s"][live], rtol=1e-10)
    assert np.all(s[~live] == 0)
    sign = np.sign(np.sum(U[:, live] * fx["U"][:, live], axis=0))
    assert_close(U[:, live] * sign, fx["U"][:, live], 1e-8, 1e-10, "U")
    assert_close(V[:, live] * sign, fx["V"][:, live], 1e-8, 1e-10, "V")


def _abs_close(a, b, rtol, atol, what):
    assert_close(np.abs(a), np.abs(b), rtol, atol, what)


def test_split_half_against_reference_and_oracle():
    from plspy_amd import split_half_resampling as sh
    from oracle import plspy_oracle as orc
    fx = load_golden("mct_split_g6x5_c3")
    X, co, S, lv = fx["X"], fx["cond_order"], fx["num_split"], fx["lv"]
    # the reference reaches the split-half loops after its perm and boot loops;
    # replay its RNG stream up to there by running the oracle's earlier phases
    np.random.seed(fx["seed"])
    smp = orc.Sampler()
    for _ in range(fx["nperm"]):
        smp.perm_task(co)
    for _ in range(fx["nboot"]):
        smp.boot(co)
    tt = sh.split_half_test_train("mct", X, None, co, S, mctype=fx["mctype"])
    res = sh.split_half("mct", X, None, co, S, mctype=fx["mctype"], lv=lv, CI=0.95)
    live = np.where(fx["s"] > 1e-10 * fx["s"].max())[0]
    nl = len(live)            # mean-centring type 0 with 2 groups: 4 of 6 are non-null
    for key in ("pls_s_train", "pls_s_train_null"):
        assert_close(tt[key][:, :nl, :], fx["tt_" + key][:, :nl, :], 1e-9, 1e-12, key)
    for key in ("pls_s_test", "pls_s_test_null"):
        _abs_close(tt[key][:nl, :nl, :], fx["tt_" + key][:nl, :nl, :], 1e-7, 1e-9, key)
        d = np.arange(nl)
        assert_close(tt[key][d, d, :], fx["tt_" + key][d, d, :], 1e-7, 1e-9, key + " diag (sign-free)")
    assert_close(np.array(tt["z"])[:nl], fx["tt_z"][:nl], 1e-6, 1e-9, "z")
    assert_close(np.array(tt["z_null"])[:nl], fx["tt_z_null"][:nl], 1e-6, 1e-9, "z_null")
    for key in ("pls_dist_u", "pls_dist_v", "pls_dist_null_u", "pls_dist_null_v"):
        _abs_close(res[key][:nl, :nl, :], fx["sh_" + key][:nl, :nl, :], 1e-7, 1e-9, key)
    for key, val in res.items():
        if not key.startswith("pls_dist"):
            assert_close(np.array(val)[:lv], fx["sh_" + key][:lv], 1e-6, 1e-9, key)


def test_full_pls_with_split_half():
    import plspy_amd
    fx = load_golden("mct_split_g6x5_c3")
    np.random.seed(fx["seed"])
    res = plspy_amd.PLS(fx["X"].copy(), fx["groups"], fx["ncond"], num_perm=fx["nperm"],
                        num_boot=fx["nboot"], mctype=fx["mctype"], pls_method="mct",
                        num_split=fx["num_split"], lv=fx["lv"])
    live = np.where(fx["s"] > 1e-10 * fx["s"].max())[0]
    assert_close(res.s[live], fx["s"][live], 1e-10, 0, "s")
    n1 = fx["nperm"] + 1
    np.testing.assert_array_equal(np.rint(res.resample_tests.permute_ratio * n1)[live],
                                  np.rint(fx["permute_ratio"] * n1)[live])
    for key in ("pls_rep_mean_u", "pls_rep_mean_v", "pls_null_mean_u", "pls_rep_z_u"):
        assert_close(np.array(res.pls_repro_sh[key]), fx["sh_" + key][: fx["lv"]], 1e-6, 1e-9, key)
    assert_close(np.array(res.pls_repro_tt["z"])[: len(live)], fx["tt_z"][: len(live)], 1e-6, 1e-9, "tt z")


# ---------------------------------------------------------------------------
# LAPACK-grade relative accuracy of the thin SVD on ill-conditioned blocks
# ---------------------------------------------------------------------------
def _hestenes_longdouble(M):
    """Singular values of M (k x p) by one-sided Jacobi in 80-bit arithmetic: the yardstick
    both LAPACK and the device are measured against (relative accuracy ~1e-18 * sqrt(p))."""
    A = np.asarray(M, dtype=np.longdouble).copy()
    k = A.shape[0]
    for _ in range(60):
        worst = 0.0
        for p in range(k - 1):
            for q in range(p + 1, k):
                a, b, g = A[p] @ A[p], A[q] @ A[q], A[p] @ A[q]
                if a == 0 or b == 0:
                    continue
                r = abs(g) / np.sqrt(a * b)
                worst = max(worst, float(r))
                if r > 1e-19:
                    tau = (b - a) / (2 * g)
                    t = (1.0 if tau >= 0 else -1.0) / (abs(tau) + np.sqrt(1 + tau * tau))
                    c = 1 / np.sqrt(1 + t * t)
                    s = t * c
                    Ap = A[p].copy()
                    A[p] = c * Ap - s * A[q]
                    A[q] = s * Ap + c * A[q]
        if worst < 1e-18:
            break
    return np.sort(np.sqrt(np.array([r @ r for r in A], dtype=np.longdouble)))[::-1].astype(float)


def _graded(k, p, ratio, seed):
    rs = np.random.RandomState(seed)
    U0, _ = np.linalg.qr(rs.randn(k, k))
    V0, _ = np.linalg.qr(rs.randn(p, k))
    sig = np.logspace(0, np.log10(ratio), k) * 37.0
    return (U0 * sig) @ V0.T


@pytest.mark.parametrize("k,ratio", [(6, 1e-3), (6, 1e-7), (38, 1e-5), (48, 1e-7), (12, 1e-5)])
def test_thin_svd_graded_spectrum_relative_accuracy(eng_factory, k, ratio):
    """north_star: "fp64 singular values match numpy.linalg.svd to 1e-10 rel".  The block handed to
    LAPACK and to the device is the SAME fp64 matrix (rows = identity), with singular values
    spread over 3 / 5 / 7 decades and randomly oriented singular vectors (the small ones are
    hidden behind cancellation, as in real data).  Every singular value -- not only the leading
    ones -- must match an 80-bit reference to 1e-10 relative, and NumPy to 1e-10 plus NumPy's own
    distance from that reference (LAPACK's dgesdd is only normwise stable: at s_k / s_1 = 1e-7 its
    values are themselves off by up to ~1e-10 relative).  Eig of a once-formed Gram (round 1) gave
    1e-8 at a ratio of 1e-4 and deflated everything below 3e-7 s_max to 0."""
    p = 5000
    M = _graded(k, p, ratio, seed=k)
    eng = eng_factory(M)
    U, s, V = eng.thin_svd(np.eye(k))
    Ul, sl, Vlt = np.linalg.svd(M, full_matrices=False)
    truth = _hestenes_longdouble(M)
    assert np.all(s > 0), "no live latent variable may be deflated"
    rel_truth = np.abs(s - truth) / truth
    lapack_err = np.abs(sl - truth) / truth
    assert rel_truth.max() < 1e-10, f"vs 80-bit reference: {rel_truth.max():.2e}"
    rel_np = np.abs(s - sl) / sl
    assert np.all(rel_np < 1e-10 + lapack_err), f"vs numpy {rel_np.max():.2e} (numpy vs reference {lapack_err.max():.2e})"
    # vectors after sign alignment: 1e-8, widened by the conditioning of each vector
    # (a perturbation of eps s_1 turns vector i by ~eps s_1 / gap_i; LAPACK's carry the same)
    sign = np.sign(np.sum(U * Ul, axis=0))
    tol = 1e-8 + 50 * np.finfo(float).eps * sl[0] / sl
    assert np.all(np.abs(U * sign - Ul).max(axis=0) < tol), np.abs(U * sign - Ul).max(axis=0) / tol
    assert np.all(np.abs(V * sign - Vlt.T).max(axis=0) < tol), np.abs(V * sign - Vlt.T).max(axis=0) / tol
    np.testing.assert_allclose(U.T @ U, np.eye(k), atol=1e-13)
    np.testing.assert_allclose(V.T @ V, np.eye(k), atol=1e-8 * max(1.0, 1e-7 / ratio))


def test_thin_svd_operator_times_data_conditioning_bound(eng_factory):
    """The same through a non-trivial operator (rows = Q, X = Q^T M): the product is now rounded
    differently on the host and on the device, which moves a singular value by up to ~eps s_1 --
    the problem's own conditioning; the comparison allows exactly that and nothing more."""
    k, p = 12, 5000
    M = _graded(k, p, 1e-6, seed=3)
    Q, _ = np.linalg.qr(np.random.RandomState(9).randn(k + 5, k + 5))
    Q = Q[:k]                                             # k x n, orthonormal rows
    X = Q.T @ M                                           # n x p
    eng = eng_factory(X)
    U, s, V = eng.thin_svd(Q)
    sl = np.linalg.svd(Q @ X, compute_uv=False)
    assert np.all(np.abs(s - sl) < 1e-10 * sl + 64 * k * np.finfo(float).eps * sl[0])


def test_thin_svd_null_space_is_deflated_at_reference_threshold(eng_factory):
    """Rank-deficient centring (mctype 0, two groups: 4 of 6 live): the null values come out as exact
    zeros with zero vectors; a REAL latent variable 1e-6 times the largest one stays live."""
    from plspy_amd import operators
    co = np.array([[10] * 3, [10] * 3])
    rs = np.random.RandomState(0)
    X = rs.randn(60, 3000)
    W = operators.mean_centre_operator(co, 0)
    U, s, V = eng_factory(X).thin_svd(W)
    sl = np.linalg.svd(W @ X, compute_uv=False)
    assert np.all(s[4:] == 0) and np.all(V[:, 4:] == 0)
    np.testing.assert_allclose(s[:4], sl[:4], rtol=1e-12)
    # a genuine but tiny latent variable: scale one cell-mean direction down by 1e-6
    M = W @ X
    Uw, sw, Vwt = np.linalg.svd(M, full_matrices=False)
    sw2 = sw.copy()
    sw2[3] = 1e-6 * sw[0]
    M2 = (Uw * sw2) @ Vwt
    U2, s2, V2 = eng_factory(M2).thin_svd(np.eye(6))
    np.testing.assert_allclose(s2[:4], np.linalg.svd(M2, compute_uv=False)[:4], rtol=1e-9)
    assert s2[3] > 0 and np.all(s2[4:] == 0)


@pytest.mark.parametrize("k", [5, 24, 48])
def test_eigh_relative_mode_on_graded_gram(eng_factory, k):
    """plsr_eigh_batch(relative = 1, init = V0): on G = D A D (A near I, D graded over 12 decades)
    every eigenvalue is relatively accurate and the returned basis is V0 @ J."""
    import torch
    rs = np.random.RandomState(k)
    eng = eng_factory(rs.randn(4, 8))
    Q, _ = np.linalg.qr(rs.randn(k + 3, k + 3))
    d = np.logspace(0, -6, k)
    B = d[:, None] * (Q[:k] + 1e-3 * rs.randn(k, k + 3))          # nearly orthogonal rows, graded norms
    G = np.zeros((1, 64, 64))
    G[0, :k, :k] = B @ B.T
    V0, _ = np.linalg.qr(rs.randn(k, k))
    ev, vec = eng.eigh(torch.as_tensor(G, device=eng.device), 0, k,
                       init=torch.as_tensor(V0[None].copy(), device=eng.device), relative=True)
    ev, vec = ev.cpu().numpy()[0], vec.cpu().numpy()[0]
    truth = _hestenes_longdouble(B) ** 2
    assert np.max(np.abs(ev - truth) / truth) < 1e-12
    J = V0.T @ vec                                                  # the pure rotation
    np.testing.assert_allclose(J.T @ J, np.eye(k), atol=1e-13)
    Gk = G[0, :k, :k]
    # (normwise only: forming J^T G J in fp64 here already costs eps |G| per entry; the relative
    # accuracy is what the eigenvalue comparison above establishes)
    np.testing.assert_allclose(J.T @ Gk @ J, np.diag(ev), atol=1e-13 * np.abs(Gk).max())
