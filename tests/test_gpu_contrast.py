"""GPU parity for the contrast variants (cst, csb, cmb) through the public
PLS() call.  U is the user's (normalised) contrast matrix, so there is no sign
or basis ambiguity: everything is compared directly with the reference's
result object."""
import numpy as np
import pytest

from tests._util import assert_close, load_golden

pytestmark = pytest.mark.gpu


def _run(fx, nboot=None):
    import plspy_amd
    kw = dict(num_perm=fx["nperm"], num_boot=fx["nboot"] if nboot is None else nboot,
              pls_method=fx["method"], contrasts=fx["contrasts_in"].copy())
    if fx["Y"] is not None:
        kw["Y"] = fx["Y"].copy()
    if fx["method"] in ("cst", "cmb"):
        kw["mctype"] = fx["mctype"]
    if fx["bscan"] is not None:
        kw["bscan"] = fx["bscan"]
    if fx["num_split"]:
        kw.update(num_split=fx["num_split"], lv=fx["lv"])
    np.random.seed(fx["seed"])
    return plspy_amd.PLS(fx["X"].copy(), fx["groups"], fx["ncond"], **kw)


@pytest.fixture(scope="module", params=["cst_g6x5_c3_q2", "csb_g6x5_c2_b2_q3", "cmb_g6x6_c3_b2_q2"])
def case(request):
    fx = load_golden(request.param)
    return fx, _run(fx)


def test_observed(case):
    fx, res = case
    assert_close(res.contrasts, fx["contrasts"], 1e-12, 1e-14, "contrasts")
    assert_close(res.s, fx["s"], 1e-10, 0, "s")
    assert_close(res.U, fx["V"], 1e-9, 1e-11, "V (voxel saliences)")       # swapped like the reference
    assert_close(res.V, fx["U"], 1e-12, 1e-14, "U (= contrasts)")
    if "R" in fx:
        assert_close(res.R, fx["R"], 1e-10, 1e-12, "R")
    if "multiblock" in fx:
        assert_close(res.multiblock, fx["multiblock"], 1e-10, 1e-12, "multiblock")
        assert_close(res.lvcorrs, fx["lvcorrs"], 1e-8, 1e-10, "lvcorrs")
    if "lvintercorrs" in fx:
        assert_close(res.lvintercorrs, fx["lvintercorrs"], 1e-9, 1e-11, "lvintercorrs")


def test_permutation(case):
    fx, res = case
    rt = res.resample_tests
    n1 = fx["nperm"] + 1
    np.testing.assert_array_equal(np.rint(rt.permute_ratio * n1), np.rint(fx["permute_ratio"] * n1))
    np.testing.assert_array_equal(np.rint(rt.stepdown_ratio * n1), np.rint(fx["stepdown_ratio"] * n1))


def test_bootstrap(case):
    fx, res = case
    if not fx["nboot"]:
        pytest.skip("the reference cannot run a csb bootstrap")
    rt = res.resample_tests
    assert_close(rt.std_errs, fx["std_errs"], 1e-8, 1e-12, "std_errs")
    assert_close(rt.boot_ratios, fx["boot_ratios"], 1e-7, 1e-9, "boot_ratios")
    assert_close(rt.conf_ints[0], fx["conf_lo"], 1e-8, 1e-11, "conf lo")
    assert_close(rt.conf_ints[1], fx["conf_hi"], 1e-8, 1e-11, "conf hi")
    if "LVcorr" in fx:
        assert_close(rt.LVcorr, fx["LVcorr"], 1e-8, 1e-11, "LVcorr")
    if "confT_lo" in fx:
        assert_close(rt.conf_ints_T[0], fx["confT_lo"], 1e-8, 1e-11, "confT lo")
        assert_close(rt.conf_ints_T[1], fx["confT_hi"], 1e-8, 1e-11, "confT hi")


def test_split_half(case):
    fx, res = case
    for tag, got in (("tt", res.pls_repro_tt), ("sh", res.pls_repro_sh)):
        for key, val in got.items():
            assert_close(np.asarray(val), fx[f"{tag}_{key}"], 1e-7, 1e-10, f"{tag}:{key}")


def test_csb_bootstrap_raises_like_the_reference():
    fx = load_golden("csb_g6x5_c2_b2_q3")
    with pytest.raises(ValueError, match="could not be broadcast"):
        _run(fx, nboot=3)
