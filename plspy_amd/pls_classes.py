"""PLS variant classes: the host side of the drop-in.

Each class keeps the reference's constructor signature, validation, attribute
names and the final U/V swap (plspy/core/pls_classes.py), so a result object
can be consumed by code written against plspy (e.g. its visualize package).
The observed decomposition (class_functions.py:98-123) runs on the device
(engine.thin_svd: Gram -> Jacobi -> back-projection), like the observed blocks and
latent scores; the permutation / bootstrap / split-half loops run on the GPU
through bootstrap_permutation.py and split_half_resampling.py."""

import numpy as np

from . import bootstrap_permutation, exceptions, operators
from . import class_functions as cf
from .engine import ProjectionEngine


METHOD_NAMES = bootstrap_permutation.METHOD_NAMES


class PLSBase:
    """Common base of the six PLS variants and their factory: ``PLSBase._create(name, X, ...)`` is what
    ``plspy_amd.PLS`` dispatches to (the reference's pls_classes.py:62-71).  VARIANTS maps the method names
    to the classes (filled in at the end of this module); an unknown name is a ValueError, a known one
    without a class exceptions.NotImplementedError."""

    VARIANTS = {}
    _pls_types = METHOD_NAMES         # (the names in messages and in __str__)

    @classmethod
    def _create(cls, pls_method, *args, **kwargs):
        variant = cls.VARIANTS.get(pls_method)
        if variant is None:
            if pls_method in METHOD_NAMES:
                raise exceptions.NotImplementedError(f"{METHOD_NAMES[pls_method]} ('{pls_method}') is not available")
            raise ValueError(f"unknown PLS method '{pls_method}' (one of {', '.join(METHOD_NAMES)})")
        return variant(*args, **kwargs)

    # helpers shared by the variants ------------------------------------
    @staticmethod
    def _get_groups_info(groups_tuple):
        """pls_classes.py:326-335."""
        if groups_tuple is None:
            return ((), 0)
        return (groups_tuple, len(groups_tuple))

    @staticmethod
    def _get_cond_order(X_shape, groups_tuple, num_conditions):
        """pls_classes.py:337-354: subjects per condition, per group."""
        if sum(groups_tuple) * num_conditions != X_shape[0]:
            raise exceptions.InputMatrixDimensionMismatchError(
                "Derived condition ordering not compatible with input matrix"
                "X's row count. Please specify a custom cond_order field.")
        return np.array([np.array([i] * num_conditions) for i in groups_tuple])

    def _take_kwargs(self, kwargs):
        """Unknown keyword arguments become attributes (pls_classes.py:201-205);
        this is how num_split, lv, bscan arrive."""
        self.pls_alg = kwargs["pls_alg"]
        self._user_defined_attrs = set()
        for key, val in kwargs.items():
            setattr(self, key, val)
            self._user_defined_attrs.add(key)

    def _resolve_cond_order(self, cond_order, groups_sizes, num_conditions, *mats):
        if cond_order is None:
            return self._get_cond_order(self.X.shape, self.groups_sizes, self.num_conditions)
        calc = sum(groups_sizes) * num_conditions
        if any(calc != m.shape[0] for m in mats):
            raise exceptions.InputMatrixDimensionMismatchError(
                "Dimension of condition orders does not match "
                "dimension of input matrix X and/or Y. Please make sure "
                "that the sum of the conditions in all groups adds "
                "up to the number of rows in the input matrices.")
        return cond_order

    def _clip_lv(self):
        """pls_classes.py:289-296 (raises AttributeError without lv, quirk Q13)."""
        max_lv = min(self.s.shape)
        if self.lv > max_lv:
            print(f"Warning: Requested lv={self.lv} exceeds maximum possible LVs ({max_lv}). "
                  f"Using lv={max_lv} instead.")
            self.lv = max_lv

    def __repr__(self):
        stg = f"\nAlgorithm: {self._pls_types[self.pls_alg]}\n\n"
        for key, val in self.__dict__.items():
            if key[0] != "_":
                stg += f"\n{key}:\n\t" + str(val).replace("\n", "\n\t")
        return stg

    __str__ = __repr__


class _MeanCentreTaskPLS(PLSBase):
    """Mean-centring task PLS (pls_classes.py:75-384)."""

    def __init__(self, X, groups_sizes, num_conditions, Y=None, cond_order=None,
                 num_perm=1000, num_boot=1000, mctype=0, CI=0.95, **kwargs):
        self._take_kwargs(kwargs)
        if len(X.shape) != 2:
            raise exceptions.ImproperShapeError("Input matrix must be 2-dimensional.")
        self.X = X
        if Y is not None:
            raise ValueError(f"Do not provide a Y/behavioural matrix for {self._pls_types[self.pls_alg]}.")
        if "contrasts" in kwargs:
            raise ValueError(f"Do not provide a contrast matrix for {self._pls_types[self.pls_alg]}.")
        self.groups_sizes, self.num_groups = self._get_groups_info(groups_sizes)
        self.num_conditions = num_conditions
        self.cond_order = self._resolve_cond_order(cond_order, groups_sizes, num_conditions, X)
        self.num_perm = num_perm
        self.num_boot = num_boot
        self.CI = CI
        if num_conditions == 1 and mctype != 1:
            print("Because you are running single condition Task PLS, "
                  "input Mean-Centering Type has to set to 1")
            self.mctype = 1
        else:
            self.mctype = mctype

        co = np.asarray(self.cond_order)
        # The upload of X (a blocking copy out of pageable memory, 2 ms at 60 x 200 000) runs in a
        # helper thread while this one draws the resampling indices (native generator on
        # np.random's state; permutation draws before bootstrap draws, as the reference's loops
        # consume them -- the observed decomposition itself draws nothing).
        import threading
        import torch
        from . import dist, resample
        box = {}
        dev = torch.cuda.current_device() if torch.cuda.is_available() else None
        cur = torch.cuda.current_stream() if dev is not None else None    # the helper thread enqueues on the caller's stream

        Wm = operators.cell_mean_operator(co)
        W = operators.mean_centre_operator(co, self.mctype)

        def upload():
            try:
                if dev is not None:
                    torch.cuda.set_device(dev)
                    torch.cuda.set_stream(cur)
                engine = box["engine"] = ProjectionEngine(X, device=None if dev is None else f"cuda:{dev}")
                # observed decomposition (pls_classes.py:258-266), on the device: the
                # two k x p blocks come from the projection kernel, the thin SVD
                # (class_functions.py:122) from the Gram + Jacobi + back-projection
                # kernels (engine.thin_svd).  Enqueued from this thread, right behind the upload (or at
                # once when X is already on the device), so that the device is busy while the
                # other thread still draws.  Nothing here waits for a result: the observed
                # blocks, the decomposition and the latent scores stay on the device for the resampling
                # phases (U for the operators, V s as the moment shift, the cell means for Tdistrib) and
                # travel to the host in page-locked buffers behind the kernels (engine.fetch_async).
                blocks = engine.apply_operator(np.vstack((Wm, W)))       # (2 cells, p) on the device
                svd = engine.thin_svd_device(W)
                Zt = engine.latents_device(svd["Vt"])                    # (X @ V)^T on the device (K5)
                box["observed"] = (blocks, svd, engine.fetch_async([blocks, svd["U"], svd["s"], svd["Vt"], Zt]))
            except BaseException as e:                       # re-raised in the caller's thread
                box["error"] = e
        th = threading.Thread(target=upload)
        th.start()
        predrawn = None
        try:
            if dist.world()[1] == 1:
                predrawn = (resample.task_permutations(co, num_perm) if num_perm > 0 else None,
                            resample.bootstraps(co, num_boot) if num_boot > 0 else None)
        finally:
            th.join()
        if "error" in box:
            raise box["error"]
        engine = box["engine"]
        blocks, svd, fetch = box["observed"]
        got = {}

        def host():
            if not got:
                b, U, sv, Vt, Z = fetch.get()
                got.update(blocks=b, U=U, s=sv, V=Vt.T, XV=np.ascontiguousarray(Z[0].T))
            return got["U"], got["s"], got["V"]

        def latent():
            host()
            return got["XV"]
        observed = dict(k=W.shape[0], U=svd["U"], VSt=svd["VSt"], Xm=blocks[:len(Wm)], host=host, XV=latent)

        self.resample_tests = bootstrap_permutation.ResampleTest._create(
            self.pls_alg, self.X, None, None, None, None, self.cond_order, self.mctype,
            preprocess=None, nperm=self.num_perm, nboot=self.num_boot, Tvsc_orig=lambda: Wm @ latent(),
            CI=self.CI, engine=engine, _observed=observed, _predrawn=predrawn)
        self.U, self.s, self.V = host()
        self.X_means, self.X_mc = got["blocks"][:len(Wm)], got["blocks"][len(Wm):]
        self.X_latent = got["XV"]                                    # X @ V (pls_classes.py:263)

        if "num_split" in self._user_defined_attrs:
            self.num_split = int(self.num_split)
            if self.num_split > 0:
                from . import split_half_resampling
                self._clip_lv()
                self.pls_repro_tt = split_half_resampling.split_half_test_train(
                    self.pls_alg, self.X, None, self.cond_order, num_split=self.num_split,
                    mctype=self.mctype, contrasts=None, engine=engine)
                self.pls_repro_sh = split_half_resampling.split_half(
                    self.pls_alg, self.X, None, self.cond_order, num_split=self.num_split,
                    mctype=self.mctype, contrasts=None, lv=self.lv, CI=self.CI, engine=engine)

        # swap U and V to be consistent with MATLAB PLS (pls_classes.py:323)
        self.U, self.V = self.V, self.U


def _check_behaviour(Y, cond_order):
    """pls_classes.py:561-564 / :1446-1449."""
    if (cf.group_stds(Y, cond_order) == 0).any():
        raise Exception("Please check your behaviour data, and make sure that none of the "
                        "columns are all the same for each group.")


class _RegularBehaviourPLS(PLSBase):
    """Regular behaviour PLS (pls_classes.py:386-647).  Defaults num_perm = 0,
    num_boot = 0 like the reference (:503-504)."""

    def __init__(self, X, groups_sizes, num_conditions, Y=None, cond_order=None,
                 num_perm=0, num_boot=0, CI=0.95, **kwargs):
        self._take_kwargs(kwargs)
        if Y is None:
            raise exceptions.MissingParameterError("Please provide a Y/behavioural matrix.")
        if "contrasts" in kwargs:
            raise ValueError(f"Do not provide a contrast matrix for {self._pls_types[self.pls_alg]}.")
        if len(X.shape) != 2 or len(Y.shape) != 2:
            raise exceptions.ImproperShapeError("Input matrices must be 2-dimensional.")
        self.X, self.Y = X, Y
        self.groups_sizes, self.num_groups = self._get_groups_info(groups_sizes)
        self.num_conditions = num_conditions
        self.cond_order = self._resolve_cond_order(cond_order, groups_sizes, num_conditions, X, Y)
        _check_behaviour(self.Y, self.cond_order)
        self.num_perm, self.num_boot, self.CI = num_perm, num_boot, CI

        engine = ProjectionEngine(X)
        co = np.asarray(self.cond_order)
        bounds = cf.cell_bounds(co)
        n = X.shape[0]
        # R = per-cell Yz.T @ Xz (class_functions.py:185-247): Xz on the device
        # (gather_zscore), the k x n behaviour operator from the tiny Y
        Xz = engine.gather_zscore(np.arange(n), bounds, np.ones(len(bounds) - 1))[0]
        eng_z = ProjectionEngine(Xz, device=engine.device, work_limit=engine.work_limit)
        A = cf.corr_operator(cf.zscore_cells(np.asarray(Y, dtype=float), bounds), bounds)
        # The k x p blocks stay on the device (R, V^T, (V s)^T) and travel to the host in page-locked
        # buffers behind the kernels: the host waits for the k x k part only (U, s: the resampling
        # operators are formed from it), the 2 x 77 MB of R and V at config 3 arrive while the tests run.
        Rd = eng_z.apply_operator(A)
        svd = eng_z.thin_svd_device(A)                                 # :574
        Zt = engine.latents_device(svd["Vt"])                          # (X @ V)^T on the device (K5)
        small = engine.fetch_async([svd["U"], svd["s"], Zt])
        large = engine.fetch_async([Rd, svd["Vt"]])
        self.U, self.s, Zt_h = small.get()
        self.X_latent = np.ascontiguousarray(Zt_h[0].T)
        self.Y_latent = cf.compute_Y_latents(self.Y, self.U, co)
        self.lvcorrs = cf.compute_corr_small(self.X_latent, self.Y, co)   # :581-583

        self.resample_tests = bootstrap_permutation.ResampleTest._create(
            self.pls_alg, self.X, self.Y, self.U, self.s, None, self.cond_order, None,
            preprocess=None, nperm=self.num_perm, nboot=self.num_boot, lvcorrs_orig=self.lvcorrs,
            CI=self.CI, engine=engine, _observed=dict(VSt=svd["VSt"]))
        self.R, Vt_h = large.get()
        self.V = Vt_h.T

        if "num_split" in self._user_defined_attrs:
            self.num_split = int(self.num_split)
            if self.num_split > 0:
                from . import split_half_resampling
                self._clip_lv()
                self.pls_repro_tt = split_half_resampling.split_half_test_train(
                    self.pls_alg, self.X, self.Y, self.cond_order, num_split=self.num_split,
                    mctype=None, contrasts=None, engine=engine)
                self.pls_repro_sh = split_half_resampling.split_half(
                    self.pls_alg, self.X, self.Y, self.cond_order, num_split=self.num_split,
                    mctype=None, contrasts=None, lv=self.lv, CI=self.CI, engine=engine)
        self.U, self.V = self.V, self.U                                 # :646


class _MultiblockPLS(PLSBase):
    """Multiblock PLS (pls_classes.py:1206-1558)."""

    def __init__(self, X, groups_sizes, num_conditions, mctype=0, Y=None, cond_order=None,
                 num_perm=1000, num_boot=1000, CI=0.95, **kwargs):
        self._take_kwargs(kwargs)
        if Y is None:
            raise exceptions.MissingParameterError("Please provide a Y/behavioural matrix.")
        if "contrasts" in kwargs:
            raise ValueError(f"Do not provide a contrast matrix for {self._pls_types[self.pls_alg]}.")
        if len(X.shape) != 2 or len(Y.shape) != 2:
            raise exceptions.ImproperShapeError("Input matrices must be 2-dimensional.")
        self.X, self.Y = X, Y
        self.groups_sizes, self.num_groups = self._get_groups_info(groups_sizes)
        self.num_conditions = num_conditions
        if num_conditions == 1 and mctype != 1:
            print("Because you are running single condition Task PLS, "
                  "input Mean-Centering Type has to set to 1")
            self.mctype = 1
        else:
            self.mctype = mctype
        self.cond_order = self._resolve_cond_order(cond_order, groups_sizes, num_conditions, X, Y)
        if "bscan" not in self._user_defined_attrs:               # :1400-1409 (0-based, quirk Q19)
            self.bscan = [i for i in range(self.num_conditions)]
        else:
            if self.bscan != sorted(self.bscan):
                print("provided bscan not in ascending order - conditions in bscan will be correctly reordered")
            if any(item < 0 or item > self.num_conditions - 1 for item in self.bscan):
                print(f"bscan should be a subset of: 1 to {self.num_conditions}")
        self.num_perm, self.num_boot, self.CI = num_perm, num_boot, CI

        co = np.asarray(self.cond_order)
        bscan = list(self.bscan)
        mask = cf.bscan_mask(co, bscan)
        self.Xbscan, self.Ybscan = self.X[mask], self.Y[mask]
        _check_behaviour(self.Ybscan, co[:, bscan])

        engine = ProjectionEngine(X)
        ng, nc = co.shape
        n = X.shape[0]
        nb, b = self.Ybscan.shape
        nbs = len(bscan)
        per = nc + nbs * b
        k = ng * per
        bounds_b = cf.cell_bounds(co[:, bscan])
        # stacked device matrix [X; bscan rows of X z-scored within cells]
        Xzb = engine.gather_zscore(np.flatnonzero(mask), bounds_b, np.ones(len(bounds_b) - 1))[0]
        import torch
        eng_c = ProjectionEngine(torch.cat((engine.X, Xzb), dim=0), device=engine.device,
                                 work_limit=engine.work_limit)
        W = operators.mean_centre_operator(co, self.mctype)
        Ab = cf.corr_operator(cf.zscore_cells(np.asarray(self.Ybscan, dtype=float), bounds_b), bounds_b)
        raw = np.zeros((k, n + nb))                                 # class_functions.py:479-511, rows per group
        for g in range(ng):
            raw[g * per:g * per + nc, :n] = W[g * nc:(g + 1) * nc]
            raw[g * per + nc:(g + 1) * per, n:] = Ab[g * nbs * b:(g + 1) * nbs * b]
        G = eng_c.fetch_async([eng_c.gram_phase(raw[None])[0]]).get()[0]
        rownorm = np.sqrt(np.diag(G)[:k])
        normed = raw / rownorm[:, None]                              # :503-505 folded into the operator
        # (as in the behaviour class: only U, s and the latent scores are waited for; the k x p
        # multiblock and V arrive in page-locked buffers while the tests run)
        Md = eng_c.apply_operator(normed)
        svd = eng_c.thin_svd_device(normed)                          # :1456
        Zt = engine.latents_device(svd["Vt"])                        # (X @ V)^T on the device (K5)
        small = engine.fetch_async([svd["U"], svd["s"], Zt])
        large = engine.fetch_async([Md, svd["Vt"]])
        self.U, self.s, Zt_h = small.get()
        XV = np.ascontiguousarray(Zt_h[0].T)                         # X @ V
        # X @ normalize(V) (:1460-1461) = (X @ V) / ||V_j||, and ||V_j|| = 1 for every live latent
        # variable, 0 (column of zeros, left alone by normalize) for a deflated one
        T_X_latent = XV
        B_X_latent = XV[mask]                                        # :1464  Xbscan @ V = (X @ V)[bscan rows]
        self.X_latent = np.vstack((T_X_latent, B_X_latent))
        self.usc, self.Tusc, self.Busc = self.X_latent, T_X_latent, B_X_latent
        Tu, Bu = cf.split_Tu_Bu(self.U, num_conditions, self.Y.shape[1], ng, nbs)
        Tusc = cf.get_Tusc(Tu, num_conditions, co)
        Busc = cf.get_Busc(Bu, self.Ybscan, co, bscan)
        Tvsc_orig = operators.cell_mean_operator(co) @ T_X_latent    # :1485
        self.Bvsc, self.Tvsc, self.Tv, self.Bv = Busc, Tusc, Tu, Bu
        self.Y_latent = np.vstack([Tusc, Busc])
        self.vsc = self.Y_latent
        self.lvcorrs = cf.compute_corr_small(B_X_latent, self.Ybscan, co[:, bscan])   # :1499-1501

        self.resample_tests = bootstrap_permutation.ResampleTest._create(
            self.pls_alg, self.X, self.Y, self.U, self.s, None, self.cond_order, self.mctype,
            preprocess=None, nperm=self.num_perm, nboot=self.num_boot, bscan=self.bscan,
            Xbscan=self.Xbscan, Ybscan=self.Ybscan, lvcorrs_orig=self.lvcorrs,
            Tvsc_orig=Tvsc_orig, CI=self.CI, engine=engine, _observed=dict(VSt=svd["VSt"]))
        self.multiblock, Vt_h = large.get()
        self.V = Vt_h.T

        if "num_split" in self._user_defined_attrs:
            self.num_split = int(self.num_split)
            if self.num_split > 0:
                from . import split_half_resampling
                self._clip_lv()
                self.pls_repro_tt = split_half_resampling.split_half_test_train(
                    self.pls_alg, self.X, self.Y, self.cond_order, num_split=self.num_split,
                    mctype=self.mctype, contrasts=None, bscan=self.bscan, Xbscan=self.Xbscan,
                    Ybscan=self.Ybscan, engine=engine)
                self.pls_repro_sh = split_half_resampling.split_half(
                    self.pls_alg, self.X, self.Y, self.cond_order, num_split=self.num_split,
                    mctype=self.mctype, contrasts=None, bscan=self.bscan, Xbscan=self.Xbscan,
                    Ybscan=self.Ybscan, lv=self.lv, CI=self.CI, engine=engine)
        self.U, self.V = self.V, self.U                              # :1557


def _contrast_decomposition(engine, rows, C):
    """class_functions.py:126-162 (_run_pls_contrast) on the device: with M =
    rows @ Z, CB = C.T @ M = (C.T rows) @ Z; U = C, s = row norms of CB,
    V = CB.T.  Returns (U, s, V)."""
    CB = engine.apply_operator(C.T @ rows).cpu().numpy()            # q x p
    return C, np.sqrt(np.sum(CB ** 2, axis=1)), CB.T


def _run_split_half(self, engine, Y, **extra):
    """The split-half block every class ends with (pls_classes.py:285-316)."""
    if "num_split" not in self._user_defined_attrs:
        return
    self.num_split = int(self.num_split)
    if self.num_split <= 0:
        return
    from . import split_half_resampling
    self._clip_lv()
    common = dict(num_split=self.num_split, mctype=getattr(self, "mctype", None),
                  contrasts=getattr(self, "contrasts", None), engine=engine, **extra)
    self.pls_repro_tt = split_half_resampling.split_half_test_train(
        self.pls_alg, self.X, Y, self.cond_order, **common)
    self.pls_repro_sh = split_half_resampling.split_half(
        self.pls_alg, self.X, Y, self.cond_order, lv=self.lv, CI=self.CI, **common)


class _ContrastTaskPLS(PLSBase):
    """Contrast task PLS (pls_classes.py:650-928)."""

    def __init__(self, X, groups_sizes, num_conditions, Y=None, cond_order=None, num_perm=1000,
                 num_boot=1000, mctype=0, contrasts=None, CI=0.95, **kwargs):
        self._take_kwargs(kwargs)
        if Y is not None:
            raise ValueError(f"Do not provide a Y/behavioural matrix for {self._pls_types[self.pls_alg]}.")
        if len(X.shape) != 2:
            raise exceptions.ImproperShapeError("Input matrix must be 2-dimensional.")
        self.X = X
        self.groups_sizes, self.num_groups = self._get_groups_info(groups_sizes)
        self.num_conditions = num_conditions
        self.cond_order = self._resolve_cond_order(cond_order, groups_sizes, num_conditions, X)
        if contrasts is None:
            raise exceptions.MissingParameterError("Please provide a contrast matrix.")
        self.contrasts = cf.normalize(np.asarray(contrasts, dtype=float))
        self.num_perm, self.num_boot, self.CI = num_perm, num_boot, CI
        self.mctype = 1 if (num_conditions == 1 and mctype != 1) else mctype

        engine = ProjectionEngine(X)
        co = np.asarray(self.cond_order)
        Wm = operators.cell_mean_operator(co)
        self.R = engine.apply_operator(Wm).cpu().numpy()                         # :853
        self.U, self.s, self.V = _contrast_decomposition(engine, Wm, self.contrasts)   # :854-856
        self.lvintercorrs = self.V.T @ self.V
        self.X_latent = engine.latents(cf.normalize(self.V))
        Tvsc_orig = Wm @ self.X_latent
        self.resample_tests = bootstrap_permutation.ResampleTest._create(
            self.pls_alg, self.X, None, self.U, self.s, self.V, self.cond_order, self.mctype,
            preprocess=None, nperm=self.num_perm, nboot=self.num_boot, contrast=self.contrasts,
            Tvsc_orig=Tvsc_orig, CI=self.CI, engine=engine)
        _run_split_half(self, engine, None)
        self.U, self.V = self.V, self.U


class _ContrastBehaviourPLS(PLSBase):
    """Contrast behaviour PLS (pls_classes.py:931-1202).  Like the reference,
    a bootstrap (num_boot > 0) ends in a broadcasting ValueError (the class hands
    the q x q lvintercorrs to the bootstrap as lvcorrs_orig, :1158)."""

    def __init__(self, X, groups_sizes, num_conditions, Y=None, cond_order=None, num_perm=1000,
                 num_boot=1000, contrasts=None, CI=0.95, **kwargs):
        self._take_kwargs(kwargs)
        if Y is None:
            raise exceptions.MissingParameterError("Please provide a Y/behavioural matrix.")
        if len(X.shape) != 2 or len(Y.shape) != 2:
            raise exceptions.ImproperShapeError("Input matrices must be 2-dimensional.")
        self.X, self.Y = X, Y
        self.groups_sizes, self.num_groups = self._get_groups_info(groups_sizes)
        self.num_conditions = num_conditions
        self.cond_order = self._resolve_cond_order(cond_order, groups_sizes, num_conditions, X, Y)
        if contrasts is None:
            raise exceptions.MissingParameterError("Please provide a contrast matrix.")
        self.contrasts = cf.normalize(np.asarray(contrasts, dtype=float))
        _check_behaviour(self.Y, self.cond_order)
        self.num_perm, self.num_boot, self.CI = num_perm, num_boot, CI

        engine = ProjectionEngine(X)
        co = np.asarray(self.cond_order)
        bounds = cf.cell_bounds(co)
        Xz = engine.gather_zscore(np.arange(X.shape[0]), bounds, np.ones(len(bounds) - 1))[0]
        eng_z = ProjectionEngine(Xz, device=engine.device, work_limit=engine.work_limit)
        A = cf.corr_operator(cf.zscore_cells(np.asarray(Y, dtype=float), bounds), bounds)
        self.R = eng_z.apply_operator(A).cpu().numpy()
        self.U, self.s, self.V = _contrast_decomposition(eng_z, A, self.contrasts)
        self.lvintercorrs = self.V.T @ self.V
        self.X_latent = engine.latents(self.V)
        self.Y_latent = cf.compute_Y_latents(self.Y, self.U, co)
        self.resample_tests = bootstrap_permutation.ResampleTest._create(
            self.pls_alg, self.X, self.Y, self.U, self.s, self.V, self.cond_order, None,
            preprocess=None, nperm=self.num_perm, nboot=self.num_boot, contrast=self.contrasts,
            lvcorrs_orig=self.lvintercorrs, CI=self.CI, engine=engine)
        _run_split_half(self, engine, self.Y)
        self.U, self.V = self.V, self.U


class _ContrastMultiblockPLS(PLSBase):
    """Contrast multiblock PLS (pls_classes.py:1561-1925)."""

    def __init__(self, X, groups_sizes, num_conditions, mctype=0, Y=None, cond_order=None,
                 num_perm=1000, num_boot=1000, contrasts=None, CI=0.95, **kwargs):
        self._take_kwargs(kwargs)
        if Y is None:
            raise exceptions.MissingParameterError("Please provide a Y/behavioural matrix.")
        if len(X.shape) != 2 or len(Y.shape) != 2:
            raise exceptions.ImproperShapeError("Input matrices must be 2-dimensional.")
        self.X, self.Y = X, Y
        self.groups_sizes, self.num_groups = self._get_groups_info(groups_sizes)
        self.num_conditions = num_conditions
        self.mctype = 1 if (num_conditions == 1 and mctype != 1) else mctype
        self.cond_order = self._resolve_cond_order(cond_order, groups_sizes, num_conditions, X, Y)
        if "bscan" not in self._user_defined_attrs:
            self.bscan = [i for i in range(self.num_conditions)]
        co = np.asarray(self.cond_order)
        bscan = list(self.bscan)
        mask = cf.bscan_mask(co, bscan)
        self.Xbscan, self.Ybscan = self.X[mask], self.Y[mask]
        _check_behaviour(self.Ybscan, co[:, bscan])
        self.num_perm, self.num_boot, self.CI = num_perm, num_boot, CI
        if contrasts is None:
            raise exceptions.MissingParameterError("Please provide a contrast matrix.")
        ng, nc = co.shape
        n = X.shape[0]
        nb, b = self.Ybscan.shape
        nbs = len(bscan)
        per = nc + nbs * b
        k = ng * per
        # keep the contrast rows of all task conditions and of the bscan behaviour rows (:1788-1799)
        Bi = np.zeros((b, nc))
        Bi[:, bscan] = 1
        keep = np.tile(np.concatenate([np.ones(nc), Bi.reshape(-1, order="F")]), ng).astype(bool)
        self.contrasts = cf.normalize(np.asarray(contrasts, dtype=float)[keep, :])

        import torch
        engine = ProjectionEngine(X)
        bounds_b = cf.cell_bounds(co[:, bscan])
        Xzb = engine.gather_zscore(np.flatnonzero(mask), bounds_b, np.ones(len(bounds_b) - 1))[0]
        eng_c = ProjectionEngine(torch.cat((engine.X, Xzb), dim=0), device=engine.device,
                                 work_limit=engine.work_limit)
        Wm = operators.cell_mean_operator(co)                        # cmb: plain cell means (class_functions.py:482)
        Ab = cf.corr_operator(cf.zscore_cells(np.asarray(self.Ybscan, dtype=float), bounds_b), bounds_b)
        raw = np.zeros((k, n + nb))
        for g in range(ng):
            raw[g * per:g * per + nc, :n] = Wm[g * nc:(g + 1) * nc]
            raw[g * per + nc:(g + 1) * per, n:] = Ab[g * nbs * b:(g + 1) * nbs * b]
        G = eng_c.gram_phase(raw[None])[0].cpu().numpy()
        normed = raw / np.sqrt(np.diag(G)[:k])[:, None]
        self.multiblock = eng_c.apply_operator(normed).cpu().numpy()
        self.U, self.s, self.V = _contrast_decomposition(eng_c, normed, self.contrasts)

        T_X_latent = engine.latents(cf.normalize(self.V))
        B_X_latent = engine.latents(self.V)[mask]
        self.X_latent = np.vstack((T_X_latent, B_X_latent))
        Tu, Bu = cf.split_Tu_Bu(self.U, num_conditions, self.Y.shape[1], ng, nbs)
        Tusc = cf.get_Tusc(Tu, num_conditions, co)
        Busc = cf.get_Busc(Bu, self.Ybscan, co, bscan)
        Tvsc_orig = Wm @ T_X_latent
        self.Bvsc, self.Tvsc, self.Tv, self.Bv = Busc, Tusc, Tu, Bu
        self.Y_latent = np.vstack([Tusc, Busc])
        self.vsc, self.usc = self.Y_latent, self.X_latent
        self.Tusc, self.Busc = T_X_latent, B_X_latent
        self.lvcorrs = cf.compute_corr_small(B_X_latent, self.Ybscan, co[:, bscan])
        self.resample_tests = bootstrap_permutation.ResampleTest._create(
            self.pls_alg, self.X, self.Y, self.U, self.s, self.V, self.cond_order, self.mctype,
            preprocess=None, nperm=self.num_perm, nboot=self.num_boot, contrast=self.contrasts,
            bscan=self.bscan, Xbscan=self.Xbscan, Ybscan=self.Ybscan, lvcorrs_orig=self.lvcorrs,
            Tvsc_orig=Tvsc_orig, CI=self.CI, engine=engine)
        _run_split_half(self, engine, self.Y, bscan=self.bscan, Xbscan=self.Xbscan, Ybscan=self.Ybscan)
        self.U, self.V = self.V, self.U


PLSBase.VARIANTS.update({"mct": _MeanCentreTaskPLS, "rb": _RegularBehaviourPLS, "mb": _MultiblockPLS, "cst": _ContrastTaskPLS, "csb": _ContrastBehaviourPLS, "cmb": _ContrastMultiblockPLS})
