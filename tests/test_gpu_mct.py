"""GPU parity for mean-centring task PLS: the HIP path (through the C ABI)
against (1) the golden vectors from the reference and (2) the NumPy oracle on
identical random draws.  Tolerances: s_hat 1e-10 rel, permutation counts exact,
bootstrap statistics 1e-9 rel on non-null latent variables."""
import numpy as np
import pytest

from oracle import plspy_oracle as orc
from tests._util import assert_close, golden_names, load_golden, nonnull, run_oracle_case

pytestmark = pytest.mark.gpu


def _run_seam(fx, **kw):
    from plspy_amd.bootstrap_permutation import ResampleTest
    X, co = fx["X"], fx["cond_order"]
    Tv = orc.group_condition_means(X @ fx["V"], co)
    np.random.seed(fx["seed"])
    return ResampleTest._create("mct", X, None, fx["U"], fx["s"].copy(), fx["V"], co, fx["mctype"],
                                nperm=fx["nperm"], nboot=fx["nboot"], Tvsc_orig=Tv,
                                keep_right_sv=True, **kw)


@pytest.fixture(scope="module", params=golden_names("mct_g"))
def case(request):
    fx = load_golden(request.param)
    return fx, _run_seam(fx)


def test_perm_against_reference(case):
    fx, rt = case
    assert_close(rt.perm_debug_dict["s_list"], fx["s_list"], 1e-10, 1e-11, "s_list")
    n1 = fx["nperm"] + 1
    np.testing.assert_array_equal(np.rint(rt.permute_ratio * n1), np.rint(fx["permute_ratio"] * n1))
    np.testing.assert_array_equal(np.rint(rt.stepdown_ratio * n1), np.rint(fx["stepdown_ratio"] * n1))


def test_boot_against_reference(case):
    fx, rt = case
    live = nonnull(fx)
    dbg = rt.boot_debug_dict
    assert_close(dbg["right_sv_sampled"], fx["right_sv_sampled"], 1e-10, 1e-12, "right_sv_sampled")
    assert_close(dbg["left_sv_sampled"], fx["left_sv_sampled"], 1e-9, 1e-12, "left_sv_sampled")
    assert_close(rt.std_errs[:, live], fx["std_errs"][:, live], 1e-9, 1e-13, "std_errs")
    assert_close(rt.boot_ratios[:, live], fx["boot_ratios"][:, live], 1e-8, 1e-10, "boot_ratios")
    assert_close(rt.conf_ints[0][:, live], fx["conf_lo"][:, live], 1e-9, 1e-12, "conf lo")
    assert_close(rt.conf_ints[1][:, live], fx["conf_hi"][:, live], 1e-9, 1e-12, "conf hi")


def test_against_oracle_same_draws(case):
    fx, rt = case
    np.random.seed(fx["seed"])
    out = run_oracle_case(fx)
    assert_close(rt.perm_debug_dict["s_list"], out["perm"]["s_list"], 1e-10, 1e-11, "s_list")
    np.testing.assert_array_equal(rt.permute_ratio, out["perm"]["permute_ratio"])
    np.testing.assert_array_equal(rt.stepdown_ratio, out["perm"]["stepdown_ratio"])
    live = nonnull(fx)
    assert_close(rt.std_errs[:, live], out["boot"]["std_errs"][:, live], 1e-9, 1e-13, "std_errs")
    assert_close(rt.boot_debug_dict["Tdistrib"][:, :, live], out["boot"]["Tdistrib"][:, :, live],
                 1e-9, 1e-12, "Tdistrib")


@pytest.mark.parametrize("name", ["mct_g10x10_c3_mc0", "mct_g3x2_c2", "mct_g8_c3_mc1"])
def test_full_pls_call(name):
    """plspy_amd.PLS(...) end to end against the reference's result object
    (per-LV sign alignment; null LVs' vectors excluded)."""
    import plspy_amd
    fx = load_golden(name)
    np.random.seed(fx["seed"])
    res = plspy_amd.PLS(fx["X"].copy(), fx["groups"], fx["ncond"], num_perm=fx["nperm"],
                        num_boot=fx["nboot"], mctype=fx["mctype"], pls_method="mct")
    live = nonnull(fx)
    # after the swap res.U is p x k (voxel saliences), res.V is k x k
    sign = np.sign(np.sum(res.V[:, live] * fx["U"][:, live], axis=0))
    assert_close(res.s[live], fx["s"][live], 1e-10, 0, "s")
    assert_close(res.V[:, live] * sign, fx["U"][:, live], 1e-8, 1e-10, "U (design saliences)")
    assert_close(res.U[:, live] * sign, fx["V"][:, live], 1e-8, 1e-10, "V (voxel saliences)")
    assert_close(res.X_mc, fx["X_mc"], 1e-10, 1e-12, "X_mc")
    assert_close(res.X_means, fx["X_means"], 1e-10, 1e-12, "X_means")
    # observed latent scores X @ V (engine.latents: K5 with a single item)
    assert_close(res.X_latent[:, live] * sign, fx["X_latent"][:, live], 1e-9, 1e-10, "X_latent")
    rt = res.resample_tests
    assert_close(rt.perm_debug_dict["s_list"][:, live], fx["s_list"][:, live], 1e-9, 1e-11, "s_list")
    n1 = fx["nperm"] + 1
    np.testing.assert_array_equal(np.rint(rt.permute_ratio * n1)[live], np.rint(fx["permute_ratio"] * n1)[live])
    assert_close(rt.std_errs[:, live], fx["std_errs"][:, live], 1e-8, 1e-12, "std_errs")
    assert_close(rt.boot_ratios[:, live] * sign, fx["boot_ratios"][:, live], 1e-7, 1e-9, "boot_ratios")


def test_ragged_and_padding_shapes():
    """Voxel counts off the 64-voxel tile, resample counts off the quad size,
    and latent-variable counts that need padding (k = 7 -> 8, k = 9 -> 12)."""
    from plspy_amd.bootstrap_permutation import ResampleTest
    rs = np.random.RandomState(0)
    for groups, nc, p, nperm, nboot in [((7,), 7, 1, 3, 5), ((3, 3, 3), 3, 65, 1, 2),
                                        ((5, 4), 1, 129, 6, 7), ((2, 2), 2, 63, 4, 4)]:
        co = np.array([[g] * nc for g in groups])
        n = co.sum()
        X = rs.randn(n, p) + rs.randn(1, p)
        mctype = 1 if nc == 1 else 0
        obs = orc.observed("mct", X, co, mctype=mctype)
        U, s, V = obs["U"], obs["s"], obs["V"]
        np.random.seed(5)
        rec = orc.RecordingSampler()
        perm = orc.permutation_test("mct", X, None, U, s, V, co, mctype, nperm, sampler=rec)
        boot = orc.bootstrap_test("mct", X, None, U, perm["s"], V, co, mctype, nboot,
                                  Tvsc_orig=obs["Tvsc_orig"], sampler=rec)
        np.random.seed(5)
        rt = ResampleTest._create("mct", X, None, U, s.copy(), V, co, mctype, nperm=nperm,
                                  nboot=nboot, Tvsc_orig=obs["Tvsc_orig"], keep_right_sv=True)
        assert_close(rt.perm_debug_dict["s_list"], perm["s_list"], 1e-10, 1e-11, f"s_list {groups}")
        assert_close(rt.boot_debug_dict["right_sv_sampled"], boot["right_sv_sampled"], 1e-10, 1e-12,
                     f"right_sv {groups}")
        live = np.where(perm["s"] > 0)[0]
        assert_close(rt.std_errs[:, live], boot["std_errs"][:, live], 1e-9, 1e-13, f"std {groups}")


def test_full_size_properties():
    """BASELINE config 2 shape (60 x 200 000) with few resamples: checks that do
    not need the oracle to run at full size.
      * sum_j s_hat_j^2 == ||W P X||_F^2, evaluated p-free through G = X X^T
        (the Gram identity of SURVEY.md H6, used here only as a cross-check);
      * std_errs on a random voxel subset == np.std of the directly computed
        projections of that subset."""
    import torch
    from plspy_amd import operators, resample
    from plspy_amd.engine import ProjectionEngine
    groups, nc, p = (10, 10), 3, 200_000
    co = np.array([[g] * nc for g in groups])
    X = np.random.RandomState(0).randn(60, p)
    W = operators.mean_centre_operator(co, 0)
    U, s, Vt = np.linalg.svd(W @ X, full_matrices=False)
    eng = ProjectionEngine(X)
    np.random.seed(1234)
    inds = resample.task_permutations(co, 12)
    ssq = eng.perm_phase(6, inds=inds, M=W.T @ U).cpu().numpy()
    G = X @ X.T
    for b in range(len(inds)):
        P = np.zeros((60, 60))
        P[np.arange(60), inds[b]] = 1
        A = W @ P
        np.testing.assert_allclose(ssq[b].sum(), np.trace(A @ G @ A.T), rtol=1e-11)
    binds = resample.bootstraps(co, 10)
    ref = Vt.T * s
    res = eng.boot_phase(6, inds=binds, M=W.T @ U, ref=ref)
    sd, _ = eng.boot_finalize(res["S1"], res["S2"], 10, num=ref)
    sub = np.random.RandomState(3).choice(p, 500, replace=False)
    sub = np.concatenate((sub, [0, 63, 64, p - 1]))
    direct = np.stack([(W @ X[binds[b]][:, sub]).T @ U for b in range(10)])
    live = s > 1e-10 * s.max()
    np.testing.assert_allclose(sd.cpu().numpy()[sub][:, live], np.std(direct, axis=0)[:, live],
                               rtol=1e-9, atol=1e-13)
    torch.cuda.synchronize()


def test_overlapped_tail_is_bit_identical():
    """plsr_set_tail_stream: running the bootstrap reductions on the tail stream
    (with a permutation phase enqueued behind the projection kernel, and the
    phase cut into several batches sharing one workspace) changes nothing."""
    import torch
    from plspy_amd import operators, resample
    from plspy_amd.engine import ProjectionEngine
    co = np.array([[6] * 3, [5] * 3])
    p = 3001
    X = np.random.RandomState(2).randn(co.sum(), p)
    W = operators.mean_centre_operator(co, 0)
    Wm = operators.cell_mean_operator(co)
    U, s, Vt = np.linalg.svd(W @ X, full_matrices=False)
    M = W.T @ U
    np.random.seed(7)
    binds = resample.bootstraps(co, 50)
    pinds = resample.task_permutations(co, 40)
    outs = []
    for overlap in (False, True):
        eng = ProjectionEngine(X, work_limit=1 << 20)          # forces several batches
        assert eng.batch_size(6, 6, 50) < 50
        Xm = eng.apply_operator(Wm)
        res = eng.boot_phase(6, inds=binds, M=M, ref=Vt.T * s, Xm=Xm, overlap_tail=overlap)
        ssq = eng.perm_phase(6, inds=pinds, M=M)
        eng.join()
        torch.cuda.synchronize()
        outs.append([t.cpu().numpy() for t in (res["S1"], res["S2"], res["ssq"], res["T"], ssq)])
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)


def test_register_resident_and_lds_fed_kernels_agree_at_full_size():
    """Config-2 shape through both projection-kernel families: n = 60 takes the
    register-resident kernels (K1r / K1br, LV-major batch order); the same data
    padded with 8 zero rows (n = 68) takes the LDS-fed kernel (quad order).  The
    zero rows change no statistic, so everything must agree to rounding."""
    from plspy_amd import operators, resample
    from plspy_amd.engine import ProjectionEngine
    groups, nc, p = (10, 10), 3, 200_000
    co = np.array([[g] * nc for g in groups])
    rs = np.random.RandomState(5)
    X = rs.randn(60, p)
    W = operators.mean_centre_operator(co, 0)
    Wm = operators.cell_mean_operator(co)
    U, s, Vt = np.linalg.svd(W @ X, full_matrices=False)
    M = W.T @ U
    ref = Vt.T * s
    np.random.seed(3)
    pinds = resample.task_permutations(co, 37)
    binds = resample.bootstraps(co, 41)
    out = []
    for pad in (0, 8):
        Xp = np.vstack([X, np.zeros((pad, p))])
        Mp = np.vstack([M, np.zeros((pad, M.shape[1]))])
        pi = np.hstack([pinds, np.tile(np.arange(60, 60 + pad, dtype=np.int32), (len(pinds), 1))])
        bi = np.hstack([binds, np.tile(np.arange(60, 60 + pad, dtype=np.int32), (len(binds), 1))])
        eng = ProjectionEngine(Xp)
        assert (eng.layout(6, 16).period == 0) == (pad == 0)
        Xm = eng.apply_operator(np.hstack([Wm, np.zeros((Wm.shape[0], pad))]))
        ssq = eng.perm_phase(6, inds=pi, M=Mp)
        res = eng.boot_phase(6, inds=bi, M=Mp, ref=ref, Xm=Xm)
        out.append([t.cpu().numpy() for t in (ssq, res["ssq"], res["T"], res["S1"], res["S2"])])
    for name, a, b in zip(("perm ssq", "boot ssq", "T", "S1", "S2"), *out):
        np.testing.assert_allclose(a, b, rtol=1e-10, atol=1e-9 * np.abs(b).max(), err_msg=name)
