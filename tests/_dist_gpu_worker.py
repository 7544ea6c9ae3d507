"""Worker of tests/test_gpu_dist.py: one rank of a two-rank (gloo) run of the real
resampling classes on the GPU; rank 0 writes what it ends with."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def problems():
    rs = np.random.RandomState(4)
    co = np.array([[5, 5], [4, 4]])
    n = int(co.sum())
    X = rs.randn(n, 700)
    Y = rs.randn(n, 3)
    return co, X, Y


def run(single):
    from oracle import plspy_oracle as orc
    from plspy_amd.bootstrap_permutation import ResampleTest
    co, X, Y = problems()
    out = {}
    obs = orc.observed("mct", X, co, mctype=0)
    np.random.seed(11)
    rt = ResampleTest._create("mct", X, None, obs["U"], obs["s"].copy(), obs["V"], co, 0, nperm=9, nboot=11,
                              Tvsc_orig=obs["Tvsc_orig"])
    out.update(mct_perm=rt.perm_debug_dict["s_list"], mct_std=rt.std_errs, mct_ratio=rt.boot_ratios,
               mct_T=rt.boot_debug_dict["Tdistrib"])
    obs = orc.observed("rb", X, co, Y=Y)
    np.random.seed(12)
    rt = ResampleTest._create("rb", X, Y, obs["U"], obs["s"].copy(), obs["V"], co, 0, nperm=7, nboot=10,
                              lvcorrs_orig=obs["lvcorrs"])
    out.update(rb_perm=rt.perm_debug_dict["s_list"], rb_std=rt.std_errs, rb_lvcorr=rt.LVcorr)
    bscan = [0, 1]
    obs = orc.observed("mb", X, co, Y=Y, mctype=0, bscan=bscan)
    np.random.seed(13)
    rt = ResampleTest._create(
        "mb", X, Y, obs["U"], obs["s"].copy(), obs["V"], co, 0, nperm=6, nboot=9, bscan=bscan,
        Xbscan=obs["Xbscan"], Ybscan=obs["Ybscan"],
        lvcorrs_orig=orc.compute_corr(obs["Xbscan"] @ obs["V"], obs["Ybscan"], co[:, bscan]),
        Tvsc_orig=orc.group_condition_means(X @ orc.normalize(obs["V"]), co))
    out.update(mb_perm=rt.perm_debug_dict["s_list"], mb_std=rt.std_errs, mb_lvcorr=rt.LVcorr,
               mb_T=rt.boot_debug_dict["Tdistrib"])
    out.update(run_split_half())
    return out


def run_split_half():
    """The sharded split-half tests (split_half_resampling._decompose: shard_bounds over the 2 S items, one
    packed all_gather of five tensors).  S = 3, 2 and 1: with two ranks the six items split evenly, with three
    ranks four items split 2 / 1 / 1 (ragged) and two items leave the last rank without any."""
    from plspy_amd import split_half_resampling as sh
    co, X, Y = problems()
    bscan = [0, 1]
    from plspy_amd import class_functions as cf
    mask = cf.bscan_mask(co, bscan)
    out = {}
    for alg, kw in (("mct", dict(mctype=0)), ("mb", dict(mctype=0, bscan=bscan, Xbscan=X[mask], Ybscan=Y[mask])),
                    ("rb", dict())):
        for S in (3, 2, 1):
            np.random.seed(20 + S)
            tt = sh.split_half_test_train(alg, X, Y if alg != "mct" else None, co, S, **kw)
            res = sh.split_half(alg, X, Y if alg != "mct" else None, co, S, lv=2, CI=0.95, **kw)
            for key in ("pls_s_train", "pls_s_test", "pls_s_train_null", "pls_s_test_null"):
                out[f"sh_{alg}_{S}_tt_{key}"] = tt[key]
            for key in ("pls_dist_u", "pls_dist_v", "pls_dist_null_u", "pls_dist_null_v"):
                out[f"sh_{alg}_{S}_{key}"] = np.abs(res[key])       # (signs of singular vectors: Jacobi's)
            out[f"sh_{alg}_{S}_rep_mean_u"] = np.array(res["pls_rep_mean_u"])
    return out


if __name__ == "__main__":
    import torch
    import torch.distributed as td
    torch.cuda.set_device(0)
    td.init_process_group("gloo")
    res = run(False)
    if td.get_rank() == 0:
        np.savez(sys.argv[1], **res)
    td.barrier()
    td.destroy_process_group()
