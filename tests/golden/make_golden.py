"""Generate golden vectors by running the REFERENCE's own plspy.core.

Run in the build container only (``/root/reference`` does not exist on the GPU
box):  ``python tests/golden/make_golden.py``.  Writes ``tests/golden/*.npz``
(data only: inputs, captured random draws, and the reference's outputs).  No
reference source travels.

Import method: ``import plspy`` fails on missing
nibabel/seaborn, so an empty stub parent package named ``plspy`` is registered
with ``__path__`` pointing at the reference tree, and ``plspy.core.pls`` is
imported beneath it.  Bytecode writing is disabled (the tree is read-only).

The reference's ``debug_dict["indices"]`` is uninitialised memory (quirk Q5), so
the random draws are captured by wrapping ``np.random.permutation`` and
``np.random.choice`` for the duration of the call.
"""
import contextlib
import importlib
import io
import os
import sys
import types

sys.dont_write_bytecode = True
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/plspy"


def load_reference():
    stub = types.ModuleType("plspy")
    stub.__path__ = [REF]
    sys.modules["plspy"] = stub
    importlib.import_module("plspy.__docs__")
    return importlib.import_module("plspy.core.pls")


class DrawRecorder:
    def __init__(self):
        self.draws = []
        self._perm = np.random.permutation
        self._choice = np.random.choice

    def __enter__(self):
        def perm(x):
            out = self._perm(x)
            self.draws.append(np.asarray(out).copy())
            return out

        def choice(*a, **k):
            out = self._choice(*a, **k)
            self.draws.append(np.asarray(out).copy())
            return out

        np.random.permutation = perm
        np.random.choice = choice
        return self

    def __exit__(self, *exc):
        np.random.permutation = self._perm
        np.random.choice = self._choice

    def packed(self):
        lens = np.array([len(d) for d in self.draws], dtype=np.int64)
        flat = np.concatenate(self.draws).astype(np.int64) if self.draws else np.zeros(0, np.int64)
        return flat, lens


def run_case(pls, name, method, n_groups, ncond, p, seed, nperm, nboot, mctype=0,
             nb=0, bscan=None, num_split=0, lv=1, data_seed=0, ncontrast=0):
    n = sum(n_groups) * ncond
    rs = np.random.RandomState(data_seed)
    X = rs.randn(n, p) + 0.5 * rs.randn(1, p)       # non-zero voxel means
    # add a group x condition effect so the leading LVs are well separated
    row = 0
    for g, ng in enumerate(n_groups):
        for c in range(ncond):
            X[row:row + ng, : p // 3] += 0.8 * (c + 1) * (1 if g % 2 == 0 else -0.5)
            X[row:row + ng, p // 3: p // 2] += 0.5 * np.sin(c + g)
            row += ng
    Y = None
    if nb:
        Y = np.random.RandomState(data_seed + 1).randn(n, nb) + 0.3 * X[:, :nb]
    kwargs = dict(num_perm=nperm, num_boot=nboot, pls_method=method)
    contrasts = None
    if ncontrast:
        # contrast rows: cst g*c; csb g*c*b; cmb the FULL g*(c + c*b) rows (the
        # class keeps the rows of the bscan conditions itself)
        rows = {"cst": len(n_groups) * ncond, "csb": len(n_groups) * ncond * nb,
                "cmb": len(n_groups) * (ncond + ncond * nb)}[method]
        contrasts = np.linalg.qr(np.random.RandomState(data_seed + 2).randn(rows, ncontrast))[0]
        kwargs["contrasts"] = contrasts.copy()
    if method in ("mct", "mb", "cst", "cmb"):
        kwargs["mctype"] = mctype
    if Y is not None:
        kwargs["Y"] = Y
    if bscan is not None:
        kwargs["bscan"] = list(bscan)
    if num_split:
        kwargs["num_split"] = num_split
        kwargs["lv"] = lv
    np.random.seed(seed)
    with DrawRecorder() as rec, contextlib.redirect_stdout(io.StringIO()), \
            np.errstate(all="ignore"):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            res = pls.PLS(X.copy(), list(n_groups), ncond, **kwargs)
    flat, lens = rec.packed()
    rt = res.resample_tests
    out = dict(
        method=np.array(method), groups=np.array(n_groups), ncond=np.array(ncond),
        mctype=np.array(mctype), seed=np.array(seed), nperm=np.array(nperm),
        nboot=np.array(nboot), num_split=np.array(num_split), lv=np.array(lv),
        bscan=np.array(bscan if bscan is not None else [], dtype=np.int64),
        X=X, draws_flat=flat, draws_len=lens,
        # the classes swap U and V at the end: res.U is the (p x k) voxel
        # saliences, res.V the (k x k) design saliences.  Store un-swapped.
        U=np.asarray(res.V), s=np.asarray(res.s), V=np.asarray(res.U),
        X_latent=np.asarray(res.X_latent),
    )
    if Y is not None:
        out["Y"] = Y
    if contrasts is not None:
        out["contrasts_in"] = contrasts
        out["contrasts"] = np.asarray(res.contrasts)
        if hasattr(res, "lvintercorrs"):
            out["lvintercorrs"] = np.asarray(res.lvintercorrs)
    if method in ("cst", "csb"):
        out["R"] = res.R
    if method == "cmb":
        out["multiblock"], out["lvcorrs"] = res.multiblock, res.lvcorrs
        out["Tusc"], out["Busc"] = res.Tusc, res.Busc
    if method == "mct":
        out["X_means"], out["X_mc"] = res.X_means, res.X_mc
    if method == "rb":
        out["R"], out["lvcorrs"] = res.R, res.lvcorrs
    if method == "mb":
        out["multiblock"], out["lvcorrs"] = res.multiblock, res.lvcorrs
        out["Tusc"], out["Busc"] = res.Tusc, res.Busc
    if nperm:
        out["permute_ratio"] = rt.permute_ratio
        out["stepdown_ratio"] = rt.stepdown_ratio
        if method in ("mct", "rb"):
            out["s_list"] = rt.perm_debug_dict["s_list"]
    if nboot:
        out["std_errs"] = rt.std_errs
        out["boot_ratios"] = rt.boot_ratios
        out["conf_lo"], out["conf_hi"] = rt.conf_ints
        out["left_sv_sampled"] = rt.boot_debug_dict["left_sv_sampled"]
        out["right_sv_sampled"] = rt.boot_debug_dict["right_sv_sampled"]
        if method in ("rb", "mb", "cmb"):
            out["LVcorr"] = rt.LVcorr
        if method in ("mb", "cmb"):
            out["confT_lo"], out["confT_hi"] = rt.conf_ints_T
    if num_split:
        for key, val in res.pls_repro_tt.items():
            out["tt_" + key] = np.asarray(val)
        for key, val in res.pls_repro_sh.items():
            out["sh_" + key] = np.asarray(val)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB, {len(lens)} draws")


def main():
    pls = load_reference()
    for mc in range(4):
        run_case(pls, f"mct_g10x10_c3_mc{mc}", "mct", (10, 10), 3, 150, 1234 + mc, 20, 20,
                 mctype=mc, data_seed=mc)
    run_case(pls, "mct_g3x2_c2", "mct", (3, 2), 2, 70, 77, 15, 15, data_seed=5)
    run_case(pls, "mct_g8_c3_mc1", "mct", (8,), 3, 65, 5, 12, 12, mctype=1, data_seed=6)
    run_case(pls, "mct_split_g6x5_c3", "mct", (6, 5), 3, 97, 99, 8, 8, num_split=8, lv=2,
             data_seed=7)
    run_case(pls, "rb_g6x5_c2_b3", "rb", (6, 5), 2, 120, 321, 12, 12, nb=3, data_seed=8)
    run_case(pls, "rb_split_g6x6_c2_b2", "rb", (6, 6), 2, 80, 11, 5, 5, nb=2, num_split=6, lv=2,
             data_seed=9)
    run_case(pls, "mb_g6x6_c3_b2", "mb", (6, 6), 3, 90, 555, 8, 8, nb=2, bscan=(1, 2),
             data_seed=10)
    run_case(pls, "mb_split_g6x5_c3_b2", "mb", (6, 5), 3, 60, 42, 4, 4, mctype=1, nb=2, bscan=(0, 2),
             num_split=5, lv=2, data_seed=11)
    # contrast variants (the reference's csb bootstrap raises a broadcast
    # ValueError, so csb fixtures carry no bootstrap)
    run_case(pls, "cst_g6x5_c3_q2", "cst", (6, 5), 3, 90, 21, 10, 10, ncontrast=2, num_split=5, lv=2,
             data_seed=12)
    run_case(pls, "csb_g6x5_c2_b2_q3", "csb", (6, 5), 2, 70, 22, 10, 0, nb=2, ncontrast=3, num_split=4, lv=2,
             data_seed=13)
    run_case(pls, "cmb_g6x6_c3_b2_q2", "cmb", (6, 6), 3, 80, 23, 8, 8, nb=2, bscan=(0, 2), ncontrast=2,
             num_split=4, lv=2, data_seed=14)


if __name__ == "__main__":
    main()
