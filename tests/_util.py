"""Shared helpers for the test-suite: golden-fixture loading and the oracle
driver that mirrors the order in which the reference's PLS() consumes random
draws (perm loop -> boot loop -> split_half_test_train -> split_half)."""
import glob
import os
import warnings

import numpy as np

from oracle import plspy_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_names(prefix=""):
    return sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        fx = {k: z[k] for k in z.files}
    fx["method"] = str(fx["method"])
    for k in ("ncond", "mctype", "seed", "nperm", "nboot", "num_split", "lv"):
        fx[k] = int(fx[k])
    fx["groups"] = [int(g) for g in fx["groups"]]
    fx["bscan"] = [int(b) for b in fx["bscan"]] if fx["bscan"].size else None
    fx["cond_order"] = np.array([[g] * fx["ncond"] for g in fx["groups"]])
    offs = np.concatenate(([0], np.cumsum(fx["draws_len"])))
    fx["draws"] = [fx["draws_flat"][offs[i]:offs[i + 1]] for i in range(len(fx["draws_len"]))]
    fx.setdefault("Y", None)
    fx.setdefault("contrasts", None)
    fx.setdefault("contrasts_in", None)
    return fx


def flat_draws(sampler_draws):
    return [np.asarray(d).ravel() for d in sampler_draws]


def run_oracle_case(fx, sampler=None, use_fixture_svd=True):
    """Run the oracle on a fixture's inputs in the reference's phase order.
    With use_fixture_svd the observed U/s/V come from the fixture so that sign
    and null-space basis are the reference's."""
    alg = fx["method"]
    co = fx["cond_order"]
    X, Y = fx["X"], fx["Y"]
    with warnings.catch_warnings(), np.errstate(all="ignore"):
        warnings.simplefilter("ignore")
        obs = orc.observed(alg, X, co, Y=Y, mctype=fx["mctype"], bscan=fx["bscan"],
                           contrasts=fx["contrasts_in"])
        if use_fixture_svd:
            U, s, V = fx["U"], fx["s"].copy(), fx["V"]
            if alg == "mct":
                obs["Tvsc_orig"] = orc.group_condition_means(X @ V, co)
            elif alg == "rb":
                obs["lvcorrs"] = orc.compute_corr(X @ V, Y, co)
            elif alg == "cst":
                obs["Tvsc_orig"] = orc.group_condition_means(X @ orc.normalize(V), co)
            elif alg == "csb":
                obs["lvcorrs"] = fx["lvintercorrs"]          # what the class passes (pls_classes.py:1158)
            else:
                obs["Tvsc_orig"] = orc.group_condition_means(X @ orc.normalize(V), co)
                obs["lvcorrs"] = orc.compute_corr(obs["Xbscan"] @ V, obs["Ybscan"], co[:, fx["bscan"]])
        else:
            U, s, V = obs["U"], obs["s"], obs["V"]
        sampler = sampler or orc.RecordingSampler()
        kw = dict(bscan=fx["bscan"], Xbscan=obs.get("Xbscan"), Ybscan=obs.get("Ybscan"), sampler=sampler,
                  contrast=fx["contrasts"])
        out = {"obs": obs}
        if fx["nperm"]:
            out["perm"] = orc.permutation_test(alg, X, Y, U, s, V, co, fx["mctype"], fx["nperm"], **kw)
            s = out["perm"]["s"]            # reference mutates s in place (Q1)
        if fx["nboot"]:
            out["boot"] = orc.bootstrap_test(alg, X, Y, U, s, V, co, fx["mctype"], fx["nboot"],
                                             lvcorrs_orig=obs.get("lvcorrs"),
                                             Tvsc_orig=obs.get("Tvsc_orig"), **kw)
        if fx["num_split"]:
            skw = dict(mctype=fx["mctype"], bscan=fx["bscan"], Ybscan=obs.get("Ybscan"),
                       lv=fx["lv"], sampler=sampler, contrasts=fx["contrasts"])
            out["tt"] = orc.split_half_both(alg, X, Y, co, fx["num_split"], which="tt", **skw)
            out["sh"] = orc.split_half_both(alg, X, Y, co, fx["num_split"], which="sh", **skw)
        out["sampler"] = sampler
    return out


def nonnull(fx, tol=1e-10):
    """Indices of latent variables whose observed singular value is not null."""
    s = np.asarray(fx["s"])
    return np.where(s > tol * max(s.max(), 1e-300))[0]


def assert_close(a, b, rtol, atol=0.0, what=""):
    """|a - b| <= atol + rtol * max(|a|, |b|) elementwise.  Non-finite entries must agree
    exactly: NaN only where the other side has NaN, infinities equal in position and sign
    (a NaN coming out of a kernel must never pass against a finite expected value)."""
    a, b = np.asarray(a, float), np.asarray(b, float)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    nan_a, nan_b = np.isnan(a), np.isnan(b)
    assert np.array_equal(nan_a, nan_b), (f"{what}: NaN mismatch ({nan_a.sum()} in result, {nan_b.sum()} expected, "
                                          f"{np.sum(nan_a != nan_b)} positions differ)")
    inf_a, inf_b = np.isinf(a), np.isinf(b)
    assert np.array_equal(inf_a, inf_b) and np.array_equal(a[inf_a], b[inf_b]), f"{what}: infinities differ"
    fin = ~(nan_a | inf_a)
    a, b = a[fin], b[fin]
    if a.size == 0:
        return
    err = np.abs(a - b)
    lim = atol + rtol * np.maximum(np.abs(a), np.abs(b))
    bad = ~(err <= lim)
    assert not bad.any(), (f"{what}: {bad.sum()} of {bad.size} off; worst abs {err.max():.3e} "
                           f"rel {np.max(err / np.maximum(np.abs(b), 1e-300)):.3e}")
