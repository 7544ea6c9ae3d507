"""BASELINE config 1 at its own shape -- mct PLS, X = 60 x 1000, groups [10, 10] x 3 conditions, 100
permutations + 100 bootstraps -- through the public seam `plspy_amd.PLS` (pls.py:21-93), against the oracle on
IDENTICAL draws: the one BASELINE shape where the oracle checks every output of every resample of a whole call
in under a second.  The reference defines this config as CPU plumbing; here the same call runs on the HIP path
(there is no other)."""
import numpy as np
import pytest

from oracle import plspy_oracle as orc
from tests._util import assert_close

pytestmark = pytest.mark.gpu


def test_config1_whole_call_every_resample():
    import plspy_amd
    groups, nc, p, nperm, nboot = [10, 10], 3, 1000, 100, 100
    co = np.array([[g] * nc for g in groups])
    X = np.random.RandomState(0).randn(60, p)
    np.random.seed(1234)
    res = plspy_amd.PLS(X, groups, nc, num_perm=nperm, num_boot=nboot, pls_method="mct")
    rt = res.resample_tests

    # ---- observed decomposition against LAPACK (oracle), per-LV sign alignment; two of the six latent
    # variables are null for any data (mctype 0: the group's cell means minus the group mean sum to zero)
    obs = orc.observed("mct", X, co, mctype=0)
    live = obs["s"] > 1e-10 * obs["s"].max()
    assert live.sum() == 4
    sign = np.sign(np.sum(res.V[:, live] * obs["U"][:, live], axis=0))
    assert_close(res.s[live], obs["s"][live], 1e-10, 0, "s")
    assert not res.s[~live].any()                                         # Q1: thresholded in place (:295)
    assert_close(res.V[:, live] * sign, obs["U"][:, live], 1e-9, 1e-11, "V (design saliences)")
    assert_close(res.U[:, live] * sign, obs["V"][:, live], 1e-9, 1e-11, "U (voxel saliences)")
    assert_close(res.X_means, obs["X_means"], 1e-12, 1e-13, "X_means")
    assert_close(res.X_mc, obs["X_mc"], 1e-11, 1e-13, "X_mc")
    assert_close(res.X_latent[:, live] * sign, (X @ obs["V"])[:, live], 1e-9, 1e-11, "X_latent")
    assert res.pls_alg == "mct" and res.num_perm == nperm and res.num_boot == nboot and res.mctype == 0
    np.testing.assert_array_equal(res.cond_order, co)

    # ---- both tests on identical draws: the oracle is given THIS call's singular vectors (their signs and
    # null-space basis are the decomposition's own) and replays np.random from the same seed
    U, s, V = res.V, res.s.copy(), res.U                                   # (swapped back, pls_classes.py:323)
    np.random.seed(1234)
    rec = orc.RecordingSampler()
    perm = orc.permutation_test("mct", X, None, U, s, V, co, 0, nperm, sampler=rec)
    Tvsc = orc.group_condition_means(X @ V, co)
    boot = orc.bootstrap_test("mct", X, None, U, perm["s"], V, co, 0, nboot, Tvsc_orig=Tvsc, sampler=rec)
    # every permutation
    np.testing.assert_array_equal(rt.perm_debug_dict["indices"], np.array(rec.draws[:nperm]).reshape(nperm, -1))
    assert_close(rt.perm_debug_dict["s_list"][:, live], perm["s_list"][:, live], 1e-10, 0, "s_hat of every permutation")
    assert np.all(rt.perm_debug_dict["s_list"][:, ~live] == 0)            # :436
    np.testing.assert_array_equal(rt.permute_ratio, perm["permute_ratio"])          # counts / (n + 1): exact
    np.testing.assert_array_equal(rt.stepdown_ratio, perm["stepdown_ratio"])
    assert_close(rt.perm_debug_dict["sum_s"], (perm["s_list"] ** 2).sum(1), 1e-10, 0, "sum_s")
    # every bootstrap
    np.testing.assert_array_equal(rt.boot_debug_dict["indices"], np.array(rec.draws[nperm:nperm + nboot]).reshape(nboot, -1))
    assert_close(rt.std_errs[:, live], boot["std_errs"][:, live], 1e-9, 0, "std_errs")
    assert_close(rt.boot_ratios[:, live], boot["boot_ratios"][:, live], 1e-9, 0, "boot_ratios")
    assert_close(rt.boot_debug_dict["Tdistrib"][:, :, live], boot["Tdistrib"][:, :, live], 1e-9, 1e-12,
                 "Tdistrib of every bootstrap")
    assert_close(rt.boot_debug_dict["left_sv_sampled"][:, :, live], boot["left_sv_sampled"][:, :, live], 1e-9, 1e-11,
                 "left_sv_sampled of every bootstrap")
    for a, b, name in zip(rt.conf_ints, boot["conf_ints"], ("lower", "upper")):
        assert_close(a[:, live], b[:, live], 1e-9, 1e-12, f"conf_ints {name}")
