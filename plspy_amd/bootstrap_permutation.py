"""Permutation and bootstrap tests on the GPU -- the drop-in for the reference's
resample seam (plspy/core/bootstrap_permutation.py:14-63, :139-263).

``ResampleTest._create(pls_alg, X, Y, U, s, V, cond_order, mctype, ...)`` keeps
the reference's signature and returns an object with the same attributes
(permute_ratio, stepdown_ratio, perm_debug_dict, conf_ints, std_errs,
boot_ratios, boot_debug_dict, ...).  What differs is how the numbers are made:

* no row gather of X and no per-iteration preprocess: resample, preprocess and
  projection onto the observed U are folded into one operator per resample
  (operators.py) which the HIP kernels contract with
  an HBM-resident matrix, a whole phase per launch.  For task PLS that matrix
  is X; for behaviour PLS it is X z-scored within cells (the permutation only
  moves Y, quirk Q15); for multiblock PLS it is [X; z-scored bscan rows];
* the bootstrap never materialises right_sv_sampled (R x p x k): moments are
  streamed in-kernel and std_errs / boot_ratios are formed once at the end;
* with torch.distributed initialised, resample ids are sharded over the ranks
  and merged with one collective per phase (dist.py).

Random draws replicate the reference's np.random call order (resample.py).

Coverage: permutation and bootstrap tests for mct, rb, mb and the contrast
variants cst / csb / cmb (projection on the normalised contrast matrix instead of
the observed U, :429-433, :658-675).  In the rb / mb bootstrap every resample
z-scores its own resampled rows, so each resample is an item of the item kernels
(rb: K4b, the two-stage kernel; mb / cmb: K4a, aggregated operators; K4f for
shapes those do not serve; then K5x / K5 for the latent scores).  The csb bootstrap ends in the reference's own ValueError
(pls_classes.py:1158 hands a q x q matrix to :725).  Nothing falls back to a CPU
path."""
import numpy as np
import torch
from scipy.stats import norm

from . import class_functions as cf
from . import dist, exceptions, operators, resample
from .engine import ProjectionEngine


# what the six method names stand for (used in messages)
METHOD_NAMES = {"mct": "mean-centred task PLS", "cst": "contrast task PLS", "rb": "behaviour PLS",
                "csb": "contrast behaviour PLS", "mb": "multiblock PLS", "cmb": "contrast multiblock PLS"}


class ResampleTest:
    """The resample seam: ``ResampleTest._create(pls_alg, X, Y, U, s, V, cond_order, mctype, ...)`` runs the
    permutation and bootstrap tests of one PLS variant and returns the object that carries their results
    (the reference's entry point, bootstrap_permutation.py:53-63, :139-159).  The interface is the method
    names and the exception types: an unknown name is a ValueError, a known one without an implementation
    exceptions.NotImplementedError.  One class serves all six variants here (_IMPLEMENTATIONS below)."""

    pls_alg = None
    _IMPLEMENTATIONS = {}             # method name -> class, filled in below the class definition

    @classmethod
    def _create(cls, pls_method, *args, **kwargs):
        impl = cls._IMPLEMENTATIONS.get(pls_method)
        if impl is None:
            if pls_method in METHOD_NAMES:
                raise exceptions.NotImplementedError(
                    f"resampling tests for {METHOD_NAMES[pls_method]} ('{pls_method}') are not available")
            raise ValueError(f"unknown PLS method '{pls_method}' (one of {', '.join(METHOD_NAMES)})")
        # the reference also leaves the method name on the class (quirk Q18: not re-entrant); the instance gets
        # it explicitly
        cls.pls_alg = pls_method
        return impl(*args, _pls_alg=pls_method, **kwargs)


def _stepdown_totals(sv):
    """totcov[r] = sum(sv[r:]**2) (bootstrap_permutation.py:317-319, :447-449)."""
    sq = np.ascontiguousarray(np.atleast_2d(sv) ** 2)
    tot = np.stack([np.sum(sq[:, r:], axis=1) for r in range(sq.shape[1])], axis=1)
    return tot if np.ndim(sv) > 1 else tot[0]


_DEGENERATE = ("Please check your behaviour data, and make sure that none of the "
               "columns are all the same for each group.")


class _ResampleTestPLS(ResampleTest):
    def __init__(self, X, Y, U, s, V, cond_order, mctype, contrast=None, preprocess=None,
                 nperm=1000, nboot=1000, bscan=None, Xbscan=None, Ybscan=None,
                 lvcorrs_orig=None, Tvsc_orig=None, CI=0.95, engine=None, keep_right_sv=False,
                 _pls_alg=None, _observed=None, _predrawn=None):
        self.pls_alg = _pls_alg or self.pls_alg
        # _observed (private, from the PLS classes): the observed decomposition as it sits on the
        # device -- dict(k, U (k, k), VSt (k, p) = (V s)^T, Xm (cells, p) cell means, host() ->
        # (U, s, V) NumPy once downloaded, XV() -> X @ V NumPy, Tvsc_orig() -> its cell means).
        # With it the phases are enqueued without waiting for U / s / V on the host; the host
        # arrays U, s, V may then be None.
        self._obs = _observed
        # _predrawn (private): (permutation, bootstrap) index tables already drawn by the caller in
        # the reference's order (while X was being uploaded)
        self._predrawn = _predrawn
        self.CI = CI
        self._cond_order = np.asarray(cond_order)
        self._mctype = mctype
        self._engine = engine if engine is not None else ProjectionEngine(X)
        self._X, self._Y = X, Y
        self._bscan, self._Ybscan = bscan, Ybscan
        n = self._engine.n
        task, behav, multi = self.pls_alg in ("mct", "cst"), self.pls_alg in ("rb", "csb"), self.pls_alg in ("mb", "cmb")
        if (self.pls_alg in ("cst", "csb", "cmb")) != (contrast is not None):
            raise exceptions.MissingParameterError("contrast variants need a contrast matrix (and only they take one)")
        # contrast variants project on the (normalised) contrast matrix instead of
        # the observed U (:429-431, :659-660) and skip the s_hat threshold
        self._C = None if contrast is None else cf.normalize(np.asarray(contrast, dtype=float))
        if self.pls_alg in ("mct", "mb"):
            if preprocess is not None and self.pls_alg == "mct":
                # a linear preprocess with the reference's signature: its operator
                # is its action on the identity
                self._W = operators.operator_from_callable(preprocess, n, self._cond_order, mctype)
            else:
                self._W = operators.mean_centre_operator(self._cond_order, mctype)
        elif self.pls_alg in ("cst", "cmb"):
            # the contrast variants use the plain cell means as task block (:389, class_functions.py:482)
            self._W = operators.cell_mean_operator(self._cond_order)

        # Task variants: the two tests are independent once their indices are drawn
        # (permutation draws first, as in the reference), so the bootstrap kernel is
        # enqueued first and its HBM-bound reduction tail overlaps the permutation
        # kernel; host post-processing follows in the reference's order.
        finish_perm = finish_boot = None
        if task:
            launch_perm = tail_boot = None
            if nperm > 0:
                launch_perm, finish_perm = self._perm_mct_start(U, s, nperm)
            if nboot > 0:
                tail_boot, finish_boot = self._bootstrap_test_start(U, s, V, nboot, Tvsc_orig, CI, keep_right_sv)
            # everything is enqueued before the host waits for anything
            if launch_perm is not None:
                launch_perm()
            if tail_boot is not None:
                tail_boot()
        if nperm > 0:
            if task:
                self.permute_ratio, self.stepdown_ratio, self.perm_debug_dict = finish_perm()
            else:
                perm = self._perm_rb if behav else self._perm_mb
                self.permute_ratio, self.stepdown_ratio, self.perm_debug_dict = perm(U, s, nperm)
        else:                                   # bootstrap_permutation.py:181-182
            self.permute_ratio = "NA"
            self.stepdown_ratio = "NA"
        if nboot > 0:
            if task:
                self.conf_ints, self.std_errs, self.boot_ratios, self.boot_debug_dict = finish_boot()
            elif self.pls_alg == "rb":           # :185-209
                (self.conf_ints, self.std_errs, self.boot_ratios, self.LVcorr,
                 self.boot_debug_dict) = self._boot_rb(U, s, V, nboot, lvcorrs_orig, CI)
            elif self.pls_alg == "csb":
                # The reference cannot finish a csb bootstrap: it is handed the q x q
                # lvintercorrs as lvcorrs_orig (pls_classes.py:1158) and subtracts a
                # (cells*b x q) array from it (:725).  Same error, before any GPU work.
                q = self._C.shape[1]
                rows = self._cond_order.size * np.asarray(Y).shape[1]
                if np.shape(lvcorrs_orig) != (rows, q) and np.shape(lvcorrs_orig) != (q,) and rows != q:
                    raise ValueError("operands could not be broadcast together with shapes "
                                     f"{np.shape(lvcorrs_orig)} ({rows},{q}) ")
                raise exceptions.NotImplementedError("csb bootstrap (the reference's own raises, see docs)")
            else:                                # :210-237
                (self.conf_ints, self.conf_ints_T, self.std_errs, self.boot_ratios, self.LVcorr,
                 self.boot_debug_dict) = self._boot_mb(U, s, V, nboot, lvcorrs_orig, Tvsc_orig, CI)
        else:                                   # :261-263
            self.conf_ints = ["NA", "NA"]
            self.std_errs = "NA"
            self.boot_ratios = "NA"
        # The result object outlives the call (the PLS result keeps it): it must not pin the engine -- X in
        # HBM, the transposed copy, gigabytes of pooled scratch -- or the device-side observed decomposition.
        # An engine made here goes away with this reference; a caller's engine gives its scratch back.
        if engine is None:
            self._engine.release_scratch()
        self._engine = None
        self._obs = None
        self._predrawn = None

    # ------------------------------------------------------------------
    # shared pieces
    # ------------------------------------------------------------------
    @staticmethod
    def _run_perm_device(eng, k, niter, inds=None, M=None, cols=None, beh=None, prepared=None):
        """Shard the phase's resamples, run the permutation kernel, gather (device tensor)."""
        rank, nranks = dist.world()
        lo, hi = dist.shard_bounds(niter, rank, nranks)
        if prepared is not None:
            ssq = eng.perm_phase(k, prepared=prepared)
        elif beh is not None:
            make, count = beh[0], hi - lo          # beh[0](a, z): Yz of this rank's resamples a..z
            ssq = eng.perm_phase(k, beh=(lambda a, z: make(lo + a, lo + z), beh[1], beh[2], count))
        elif cols is not None:
            ssq = eng.perm_phase(k, cols=cols[lo:hi])
        else:
            ssq = eng.perm_phase(k, inds=inds[lo:hi], M=M)
        (ssq,), _ = dist.exchange([ssq], [], niter)
        return ssq

    @classmethod
    def _run_perm(cls, eng, k, niter, **kw):
        return cls._run_perm_device(eng, k, niter, **kw).cpu().numpy()

    @staticmethod
    def _ratios(s_hat, s_ref, step_ref, niter):
        greatersum = np.sum(s_hat >= s_ref, axis=0).astype(float)                 # :427/:437
        step = np.sum(_stepdown_totals(s_hat) >= _stepdown_totals(step_ref), axis=0).astype(float)
        return greatersum / (niter + 1), step / (niter + 1)                       # :444, :452 (Q2)

    def _draw_on_rank0(self, fn):
        rank, _ = dist.world()
        out = fn() if rank == 0 else None
        return dist.broadcast_indices(out, self._engine.device)

    # ------------------------------------------------------------------
    # permutation tests
    # ------------------------------------------------------------------
    def _perm_mct_start(self, U, s, niter, threshold=1e-12):
        """bootstrap_permutation.py:266-464 for mct and cst (cst: cell means
        projected on the normalised contrasts, no threshold on s_hat, :429-433).
        Draws the indices now; returns (launch, finish): launch() enqueues the kernel and the
        download of its result, finish() waits for it and does the host summary."""
        eng = self._engine
        obs = self._obs if self._C is None else None
        if obs is not None:
            k = obs["k"]
            # M = W^T U (n x k) where U lives:  (U^T W)^T, a layout copy of k x n numbers
            Wd = eng.dev(np.ascontiguousarray(self._W, dtype=np.float64))
            M = eng.rotate_rows(obs["U"][None], Wd[None])[0].t().contiguous()
        else:
            Uq = np.asarray(U, dtype=float) if self._C is None else self._C
            k = Uq.shape[1]
            s[np.abs(s) < threshold] = 0        # in place, like the reference (:295, quirk Q1)
            M = self._W.T @ Uq                  # n x k:  VS = X^T (P^T W^T U)
        if self._predrawn is not None and self._predrawn[0] is not None:
            inds = self._predrawn[0]
        else:
            inds = self._draw_on_rank0(lambda: resample.task_permutations(self._cond_order, niter))
        pending = []
        # the operator fragments are built now, on a side stream: the bootstrap phase that is enqueued between
        # this and launch() then does not wait for them
        rank, nranks = dist.world()
        lo, hi = dist.shard_bounds(niter, rank, nranks)
        prepared = eng.perm_prepare(k, inds[lo:hi], M)

        def launch():
            pending.append(eng.fetch_async([self._run_perm_device(eng, k, niter, inds=inds, M=M, prepared=prepared)]))

        def finish():
            s_obs = s
            if obs is not None:
                _, s_obs, _ = obs["host"]()
                s_obs[np.abs(s_obs) < threshold] = 0        # (:295, Q1) on the array the caller keeps
            s_hat = np.sqrt(pending[0].get()[0])
            if self._C is None:
                s_hat[np.abs(s_hat) < threshold] = 0    # :436
            ratio, step = self._ratios(s_hat, s_obs, np.copy(s_obs), niter)
            total = np.sum(s_hat ** 2, axis=1)
            debug = {
                "s_list": s_hat,
                # U is square orthogonal, so sum(permuted**2) == sum(s_hat**2); the
                # reference stores the two under swapped keys (quirk Q5)
                "sum_s": total,
                "sum_perm": total.copy(),
                "indices": inds,
            }
            return ratio, step, debug
        return launch, finish

    def _perm_mct(self, U, s, niter, threshold=1e-12):
        launch, finish = self._perm_mct_start(U, s, niter, threshold)
        launch()
        return finish()

    def _draw_behaviour_perms(self, Ysrc, niter, with_task):
        """Row permutations of the behaviour block (and, for multiblock, the task
        permutation drawn just before each), redrawn up to 100 times while any
        group std of the permuted Y is 0 (:333-355; the guard uses the full
        cond_order even on the bscan subset, quirk Q8)."""
        co = self._cond_order
        nrows = Ysrc.shape[0]
        is_bad = cf.degenerate_guard(Ysrc, co)
        if with_task:
            got = resample.draw_guarded(
                niter, lambda m: resample.mb_permutation_tries(co, nrows, m),         # :343, :347
                lambda task, rows: is_bad(rows))
        else:
            got = resample.draw_guarded(
                niter, lambda m: (resample.permutations(nrows, m),),                  # :338
                is_bad)
        if got is None:
            raise Exception(_DEGENERATE)                                               # :355
        return np.concatenate(got, axis=1) if with_task else got[0]

    def _perm_rb(self, U, s, niter, threshold=1e-12):
        """:266-464 for rb.  Only Y is permuted; X enters through its per-cell
        z-scores, computed once on the device (the reference recomputes them in
        every iteration, quirk Q15)."""
        eng = self._engine
        co = self._cond_order
        Y = np.asarray(self._Y, dtype=float)
        U = np.asarray(U, dtype=float) if self._C is None else self._C     # csb: contrasts (:429-431)
        n, b = Y.shape
        k = U.shape[1]
        bounds = cf.cell_bounds(co)
        s[np.abs(s) < threshold] = 0
        perms = self._draw_on_rank0(lambda: self._draw_behaviour_perms(Y, niter, False))
        Xz = eng.gather_zscore(np.arange(n), bounds, np.ones(len(bounds) - 1))[0]
        eng_z = ProjectionEngine(Xz, device=eng.device, work_limit=eng.work_limit)
        # operator column (b, j)[i] = sum_beh Yz_b[i, beh] * U[(cell(i), beh), j]
        # (the R x k x n operators are formed on the device from Yz and U)
        # and the z-scored behaviour itself is made batch by batch, behind the kernels)
        rowcell = np.repeat(np.arange(len(bounds) - 1), np.diff(bounds)).astype(np.int32)
        s_hat = np.sqrt(self._run_perm(
            eng_z, k, niter, beh=(lambda a, z: cf.zscore_cells(Y[perms[a:z]], bounds), U, rowcell)))
        if self._C is None:
            s_hat[np.abs(s_hat) < threshold] = 0
        ratio, step = self._ratios(s_hat, s, np.copy(s), niter)
        total = np.sum(s_hat ** 2, axis=1)
        debug = {"s_list": s_hat, "sum_s": total, "sum_perm": total.copy(), "indices": perms}
        return ratio, step, debug

    def _mb_cell_layout(self):
        """The multiblock of a resample as cells of gathered rows (engine.split_gram / split_rows): behaviour cells
        first (group x bscan condition), then the task cells (group x condition), whose sums give the task rows --
        possible when the mean-centring operator W is constant inside a cell.  Returns the constant part of the
        cell description, or None."""
        co = self._cond_order
        bscan = list(self._bscan)
        ng, nc = co.shape
        b = np.asarray(self._Ybscan).shape[1]
        nbs = len(bscan)
        per = nc + nbs * b
        bounds_b, bounds_t = cf.cell_bounds(co[:, bscan]), cf.cell_bounds(co)
        W = self._W
        Wcell = W[:, bounds_t[:-1]]
        if not np.array_equal(W, np.repeat(Wcell, np.diff(bounds_t), axis=1)):
            return None
        row_cell, row_sub = [], []
        for g in range(ng):
            for r in range(per):
                row_cell.append(-1 if r < nc else g * nbs + (r - nc) // b)
                row_sub.append(g * nc + r if r < nc else (r - nc) % b)
        ncb = len(bounds_b) - 1
        return dict(cell_rows=[int(x) for x in np.diff(bounds_b)] + [int(x) for x in np.diff(bounds_t)], nbq=ncb,
                    Wc=np.concatenate((np.zeros((ng * nc, ncb)), Wcell), axis=1), row_cell=row_cell, row_sub=row_sub)

    def _perm_mb_grams(self, U, k, niter, task_perm, beh_perm):
        """The multiblock permutation from per-resample Grams (K2s, engine.split_gram): with R_r the un-normalised
        rows of resample r (k x p) and D_r their norms, the statistic sum_v (U^T D_r^-1 R_r)^2 per latent variable is
        diag(U^T Gn_r U) with Gn_r = D_r^-1 R_r R_r^T D_r^-1 -- one pass over the voxels per resample, rows of X read by
        index, instead of two dense k x (n + nb) products.  Returns (total variance of the observed block, row
        norms^2 (niter, k), s_hat^2 (niter, q)) or None when the kernel does not serve the shape."""
        import torch
        eng = self._engine
        layout = self._mb_cell_layout()
        if layout is None:
            return None
        n = int(self._cond_order.sum())
        Yb = np.asarray(self._Ybscan, dtype=float)
        nb = Yb.shape[0]
        brows = np.flatnonzero(cf.bscan_mask(self._cond_order, list(self._bscan)))

        def grams(task, beh):
            cells = dict(layout, normalise=True,
                         xsrc=np.concatenate((np.broadcast_to(brows, beh.shape), task), axis=1),
                         ysrc=np.concatenate((beh, np.zeros_like(task)), axis=1))
            return eng.split_gram(cells, Yb)
        got = grams(np.arange(n, dtype=np.int32)[None], np.arange(nb, dtype=np.int32)[None])
        if got is None:
            return None
        total_s = float((got[1][0, :k] ** 2).sum().item())
        rank, nranks = dist.world()
        lo, hi = dist.shard_bounds(niter, rank, nranks)
        G, rown = grams(task_perm[lo:hi], beh_perm[lo:hi]) if hi > lo else \
            (torch.zeros((0, k, k), dtype=torch.float64, device=eng.device),) * 2
        Ut = eng.dev(np.ascontiguousarray(U.T, dtype=np.float64))                 # (q, k)
        ssq = ((Ut @ G[:, :k, :k]) * Ut).sum(-1).clamp_(min=0.0) if hi > lo else G.new_zeros((0, U.shape[1]))
        rn2 = rown[:, :k] ** 2 if hi > lo else G.new_zeros((0, k))
        (ssq, rn2), _ = dist.exchange([ssq.contiguous(), rn2.contiguous()], [], niter)
        return total_s, rn2.cpu().numpy(), ssq.cpu().numpy()

    def _perm_mb(self, U, s, niter, threshold=1e-12):
        """:266-464 for mb.  The multiblock rows are linear in the stacked matrix
        [X; z-scored bscan rows of X]; the per-row normalisation over all voxels
        (class_functions.py:503-505) is folded into the projection as
        raw.T @ (D^-1 U) after a first pass that returns the row norms D
        (SURVEY.md appendix A6)."""
        eng = self._engine
        co = self._cond_order
        bscan = list(self._bscan)
        Yb = np.asarray(self._Ybscan, dtype=float)
        U = np.asarray(U, dtype=float) if self._C is None else self._C     # cmb: contrasts
        ng, nc = co.shape
        n = int(co.sum())
        nb, b = Yb.shape
        q = U.shape[1]                                       # latent variables (mb: = k)
        nbs = len(bscan)
        per = nc + nbs * b                                   # rows per group in the stacked block
        k = ng * per                                         # rows of the multiblock
        s[np.abs(s) < threshold] = 0
        mask = cf.bscan_mask(co, bscan)
        bounds_b = cf.cell_bounds(co[:, bscan])
        draws = self._draw_on_rank0(lambda: self._draw_behaviour_perms(Yb, niter, True))
        task_perm, beh_perm = draws[:, :n], draws[:, n:]
        fast = self._perm_mb_grams(U, k, niter, task_perm, beh_perm)
        if fast is not None:
            total_s, rownorm2, ssq = fast
            org_s = np.sqrt(s ** 2 / np.sum(s ** 2) * total_s)
            total_hat = rownorm2.sum(axis=1)                                         # :419
            s_hat = np.sqrt(ssq)                                                      # :404-405 / :431-432
            if self._C is None:
                per_hat = s_hat ** 4 / np.sum(s_hat ** 4, axis=1, keepdims=True)      # quirk Q3 (:421-423)
                s_hat = np.sqrt(per_hat * total_hat[:, None])                         # :424
                ratio, step = self._ratios(s_hat, org_s, org_s, niter)
            else:
                # cmb compares with s itself (:433) but steps down against the rescaled org_s (:312-319)
                ratio, step = self._ratios(s_hat, s, org_s, niter)
            return ratio, step, {"s_list": s_hat, "indices": draws, "org_s": org_s}
        Xzb = eng.gather_zscore(np.flatnonzero(mask), bounds_b, np.ones(len(bounds_b) - 1))[0]
        eng_c = ProjectionEngine(torch.cat((eng.X, Xzb), dim=0), device=eng.device,
                                 work_limit=eng.work_limit)

        def raw_rows(task_rows, Yz):
            """k x (n + nb) operators of the un-normalised multiblock rows for a
            batch: task_rows (R, g*c, n), Yz (R, nb, b) z-scored behaviour."""
            R = task_rows.shape[0]
            A = np.zeros((R, k, n + nb))
            for g in range(ng):
                A[:, g * per:g * per + nc, :n] = task_rows[:, g * nc:(g + 1) * nc]
                for ci in range(nbs):
                    lo, hi = bounds_b[g * nbs + ci], bounds_b[g * nbs + ci + 1]
                    r0 = g * per + nc + ci * b
                    A[:, r0:r0 + b, n + lo:n + hi] = np.transpose(Yz[:, lo:hi], (0, 2, 1))
            return A

        # observed un-normalised block: total variance and the rescaled s (:305-312)
        A0 = raw_rows(self._W[None], cf.zscore_cells(Yb, bounds_b)[None])
        total_s = self._run_perm(eng_c, k, 1, cols=A0)[0].sum()
        org_s = np.sqrt(s ** 2 / np.sum(s ** 2) * total_s)

        # task rows of resample r: W P_r  (column i collects W's columns r with perm[r] == i)
        task_rows = np.zeros((niter, ng * nc, n))
        ridx = np.arange(niter)[:, None]
        task_rows[ridx, :, task_perm] = self._W.T[None]      # [r, :, perm[r, q]] = W[:, q]
        A = raw_rows(task_rows, cf.zscore_cells(Yb[beh_perm], bounds_b))
        rownorm2 = self._run_perm(eng_c, k, niter, cols=A)                       # R x k row norms^2
        total_hat = rownorm2.sum(axis=1)                                         # :419
        with np.errstate(divide="ignore", invalid="ignore"):
            scaledU = U[None] / np.sqrt(rownorm2)[:, :, None]                    # D^-1 U per resample
        cols = np.swapaxes(scaledU, 1, 2) @ A                         # (r,j,k)(r,k,i) -> r,j,i
        s_hat = np.sqrt(self._run_perm(eng_c, q, niter, cols=cols))              # :404-405 / :431-432
        if self._C is None:
            per_hat = s_hat ** 4 / np.sum(s_hat ** 4, axis=1, keepdims=True)      # quirk Q3 (:421-423)
            s_hat = np.sqrt(per_hat * total_hat[:, None])                         # :424
            ratio, step = self._ratios(s_hat, org_s, org_s, niter)
        else:
            # cmb compares with s itself (:433) but steps down against the rescaled org_s (:312-319)
            ratio, step = self._ratios(s_hat, s, org_s, niter)
        debug = {"s_list": s_hat, "indices": draws, "org_s": org_s}
        return ratio, step, debug

    # ------------------------------------------------------------------
    # bootstrap test (mct)
    # ------------------------------------------------------------------
    def _bootstrap_test(self, U, s, V, niter, Tvsc_orig, CI, keep_right_sv):
        tail, finish = self._bootstrap_test_start(U, s, V, niter, Tvsc_orig, CI, keep_right_sv)
        tail()
        return finish()

    def _bootstrap_test_start(self, U, s, V, niter, Tvsc_orig, CI, keep_right_sv):
        """bootstrap_permutation.py:467-766 for mct and cst, streaming form.  cst:
        the projection is on the normalised contrasts (VS = permuted.T @ C, :620
        with U = C) and boot_ratios = V / std_errs (:703).  Draws the indices and
        enqueues the kernels now (reductions on the engine's tail stream).  Returns (tail,
        finish): tail() enqueues the exchange, the final statistics and their download (call it
        after whatever should overlap the reductions has been enqueued), finish() waits and
        does the host summary."""
        eng = self._engine
        co = self._cond_order
        obs = self._obs if self._C is None else None
        rank, nranks = dist.world()
        if self._predrawn is not None and self._predrawn[1] is not None:
            inds = self._predrawn[1]
        else:
            inds = self._draw_on_rank0(lambda: resample.bootstraps(co, niter))
        lo, hi = dist.shard_bounds(niter, rank, nranks)
        Wm = operators.cell_mean_operator(co)
        if obs is not None:
            k = obs["k"]
            Wd = eng.dev(np.ascontiguousarray(self._W, dtype=np.float64))
            M = eng.rotate_rows(obs["U"][None], Wd[None])[0].t().contiguous()     # W^T U, on the device
            ref = obs["VSt"].t().contiguous()            # observed VS = V s (p x k): a layout copy
            Xm = obs["Xm"]
        else:
            U = np.asarray(U, dtype=float) if self._C is None else self._C
            k = U.shape[1]
            V = np.asarray(V, dtype=float)
            M = self._W.T @ U
            Xm = eng.apply_operator(Wm)                   # k x p cell means of X, on device
            # observed VS (shift of the moment sums and numerator of the ratios):
            # X_mc.T @ U = V s for mct; R.T @ C = V for cst
            ref = eng.dev(V * s if self._C is None else V)
        res = eng.boot_phase(k, inds=inds[lo:hi], M=M, ref=ref, Xm=Xm, dump=keep_right_sv,
                             overlap_tail=True)
        pending = []

        early = {}

        def tail():
            # exchange and final statistics on the tail stream, behind the reductions: they proceed while
            # the main stream runs whatever was enqueued after the bootstrap kernel (the permutation)
            per = [res["ssq"], res["T"]] + ([res["vs"]] if keep_right_sv else [])
            with eng.tail_stream():
                per, (S12,) = dist.exchange(per, [res["S12"]], niter)
                sd, ratio = eng.boot_finalize(S12[0], S12[1], niter, num=ref)  # :695, :701
            eng.join()
            for t in [sd, ratio, S12] + list(per):
                t.record_stream(torch.cuda.current_stream())
            pending.append(eng.fetch_async([sd, ratio] + list(per)))
            if obs is not None:
                # the p-free part of the host summary needs the observed X @ V only: formed here,
                # while the device runs the kernels enqueued above, not after the last download
                early["left"] = self._W @ obs["XV"]()[inds]

        def finish():
            Vh = V
            if obs is not None:
                _, _, Vh = obs["host"]()
            got = pending[0].get()
            return self._bootstrap_test_finish(got[0], got[1], got[2:], inds, Vh, Tvsc_orig, CI, keep_right_sv,
                                               left=early.get("left"))
        return tail, finish

    def _bootstrap_test_finish(self, std_errs, boot_ratios, per, inds, V, Tvsc_orig, CI, keep_right_sv,
                               left=None):
        eng = self._engine
        # Tdistrib[i] = cell means of X @ normalize(VS_i)  (:623, :633-634)
        norms = np.sqrt(per[0])                                       # R x k
        T = per[1]                                                    # R x k(lv) x k(cell)
        with np.errstate(divide="ignore", invalid="ignore"):
            Td = np.where(norms[:, :, None] != 0, T / norms[:, :, None], 0.0)
        Tdistrib = np.transpose(Td, (0, 2, 1))                        # R x cell x lv
        z = norm.ppf(1 - (1 - CI) / 2)                                # :709
        half = np.std(Tdistrib, axis=0) * z                           # :715-716
        if callable(Tvsc_orig):
            Tvsc_orig = Tvsc_orig()
        conf_int = (Tvsc_orig - half, Tvsc_orig + half)               # :717

        # left_sv_sampled[i] = permuted_i @ V = W P_i (X V)   (:617, :631) -- p-free
        if left is None:
            if self._obs is not None and self._C is None:
                XV = self._obs["XV"]()                                # already formed for X_latent
            else:
                XV = eng.latents(V)                                   # X @ V on the device (K5, one item)
            left = self._W @ XV[inds]                                 # (c,r)(b,r,k) -> b,c,k
        debug = {
            "left_sv_sampled": left,
            "right_sv_sampled": per[2] if keep_right_sv else None,
            "indices": inds,
            "Tdistrib": Tdistrib,
        }
        return conf_int, std_errs, boot_ratios, debug

    # ------------------------------------------------------------------
    # bootstrap tests (rb, mb): every resample z-scores its own rows
    # ------------------------------------------------------------------
    def _draw_boot_with_guard(self, niter, Ysrc, multiblock):
        """Bootstrap index vectors with the reference's degenerate-Y redraw
        (:542-572): up to 100 draws per iteration while any group std of the
        resampled Y is 0 (guard on the full cond_order, quirk Q8).  rb: one draw
        per try; mb: task draw then bscan draw per try (:547-553, quirk Q7)."""
        co = self._cond_order
        is_bad = cf.degenerate_guard(Ysrc, co)                                          # :563-564
        if multiblock:
            got = resample.draw_guarded(
                niter, lambda m: resample.mb_bootstrap_tries(co, self._bscan, m),     # :547, :551
                lambda ti, bi: is_bad(bi))
        else:
            got = resample.draw_guarded(
                niter, lambda m: (resample.bootstraps(co, m),),                       # :557
                is_bad)
        if got is None:
            raise Exception(_DEGENERATE)                                               # :572
        return np.concatenate(got, axis=1) if multiblock else got[0]

    def _observed_vs(self, V, s):
        """The observed V s (p x k) on the device, as a function to call once it is needed: a layout
        copy of the (V s)^T the PLS classes left there (_observed), or formed from the host arrays
        and uploaded in the background."""
        if self._obs is not None and "VSt" in self._obs:
            ref = self._obs["VSt"].t().contiguous()
            return lambda: ref
        eng = self._engine
        return self._upload_in_background(lambda: eng.scale_cols(V, s), upload=False)

    def _upload_in_background(self, make, upload=True):
        """Form a large host array (the observed V s, p x k) and upload it from a helper thread while
        the caller draws the bootstrap indices and runs the degenerate-Y guard (both single-threaded
        host work, 8 + 11 ms at config 3).  Returns a function that joins and hands over the
        device tensor."""
        import threading
        import torch
        eng = self._engine
        box = {}
        dev, cur = torch.cuda.current_device(), torch.cuda.current_stream()

        def work():
            try:
                torch.cuda.set_device(dev)
                torch.cuda.set_stream(cur)
                box["t"] = eng.dev(make()) if upload else make()
            except BaseException as e:                       # re-raised in the caller's thread
                box["error"] = e
        th = threading.Thread(target=work)
        th.start()

        def get():
            th.join()
            if "error" in box:
                raise box["error"]
            return box["t"]
        return get

    class _RunningStd:
        """np.std(axis=0) of per-resample rows that arrive batch by batch: sums about the first row,
        accumulated while the device works on the next batch instead of one pass over the whole
        (niter, rows, k) array after the last one (6 ms at config 3).  The shift keeps the one-pass
        formula at a few ulp for these O(1) correlations / cell means."""

        def __init__(self):
            self.n, self.shift, self.s1, self.s2 = 0, None, 0.0, 0.0

        def add(self, rows):
            if self.shift is None:
                self.shift = rows[0].copy()
            d = rows - self.shift
            self.s1 = self.s1 + d.sum(axis=0)
            self.s2 = self.s2 + np.einsum("i...,i...->...", d, d)
            self.n += len(rows)

        def std(self):
            m = self.s1 / self.n
            return np.sqrt(np.maximum(self.s2 / self.n - m * m, 0.0))

    @staticmethod
    def _normalised_latents(zt, nsq):
        """X @ normalize(VS_b) (:623) for a batch, from (X VS_b)^T (cnt, k, n) and the
        squared column norms of VS_b (cnt, k): (cnt, n, k)."""
        Z = np.transpose(zt, (0, 2, 1))
        norms = np.sqrt(nsq)
        with np.errstate(divide="ignore", invalid="ignore"):
            return np.where(norms[:, None, :] != 0, Z / norms[:, None, :], 0.0)

    def _finalize_early(self, niter, ref):
        """boot_items callback (single rank): std_errs / boot_ratios and their download, enqueued behind
        the last batch so that they overlap the host's consumption of the pending batches."""
        _, nranks = dist.world()
        if nranks > 1:
            return None
        eng = self._engine
        return lambda S1, S2: eng.fetch_async(list(eng.boot_finalize(S1, S2, niter, num=ref)))

    def _finish_items(self, res, per_resample, niter, ref):
        """Exchange a sharded boot_items result (moment sums, and the per-resample
        summaries the host formed batch by batch) and form std_errs / boot_ratios."""
        eng = self._engine
        _, nranks = dist.world()
        if nranks > 1:
            full, (S12,) = dist.exchange([eng.dev(a) for a in per_resample], [res["S12"]], niter)
            per_resample = [t.cpu().numpy() for t in full]
            S1, S2 = S12[0], S12[1]
        else:
            S1, S2 = res["S1"], res["S2"]
        if res.get("after") is not None and nranks == 1:
            sd_h, ratio_h = res["after"].get()                                 # enqueued behind the last batch
            return sd_h, ratio_h, per_resample
        sd, ratio = eng.boot_finalize(S1, S2, niter, num=ref)                  # :695, :701
        sd_h, ratio_h = eng.fetch_async([sd, ratio]).get()                     # (page-locked buffers, see engine)
        return sd_h, ratio_h, per_resample

    def _boot_rb(self, U, s, V, niter, lvcorrs_orig, CI):
        """bootstrap_permutation.py:467-766 for rb."""
        eng = self._engine
        co = self._cond_order
        Y = np.asarray(self._Y, dtype=float)
        U = np.asarray(U, dtype=float)
        V = None if V is None else np.asarray(V, dtype=float)
        n, b = Y.shape
        k = U.shape[1]
        bounds = cf.cell_bounds(co)
        ref_ready = self._observed_vs(V, s)
        inds = self._draw_on_rank0(lambda: self._draw_boot_with_guard(niter, Y, False))
        rank, nranks = dist.world()
        lo, hi = dist.shard_bounds(niter, rank, nranks)
        mine = inds[lo:hi]
        ref = ref_ready()

        def ops_fn(a, z, _):
            # op_b[j, i] = sum_beh Yz_b[i, beh] U[(cell(i), beh), j]:  VS_b = R_b^T U with
            # R_b = Yz_b^T Xz_b per cell (class_functions.py:240-242, :620)
            Yz = cf.zscore_cells(Y[mine[a:z]], bounds)
            ops = np.empty((z - a, k, n))
            for c, (l, h) in enumerate(zip(bounds[:-1], bounds[1:])):
                ops[:, :, l:h] = np.swapaxes(Yz[:, l:h] @ U[c * b:(c + 1) * b], 1, 2)
            return ops

        LVc = np.empty((hi - lo, (len(bounds) - 1) * b, k))
        spread = self._RunningStd()
        yz_of = {}                                   # a batch's z-scored behaviour: formed once, used twice

        def yz(a, z):
            if (a, z) not in yz_of:
                yz_of[a, z] = cf.zscore_cells(Y[mine[a:z]], bounds)
            return yz_of[a, z]

        def on_batch(a, z, zt, nsq, form):
            # LVcorr_b = _compute_corr(X_new @ V_hat, Y_new)   (:638-641); X_new @ V_hat = (X @ V_hat)[inds]
            # (no transposes, and no division by the column norms: see lvcorr_from_latents)
            # -- the engine computes exactly those columns (latent_index: the sample's rows)
            LVc[a:z] = cf.lvcorr_from_latents(zt, yz(a, z), bounds)
            yz_of.pop((a, z), None)
            spread.add(LVc[a:z])

        res = eng.boot_items(mine, bounds, np.ones(len(bounds) - 1), k, ops_fn, ref=ref, on_batch=on_batch,
                             beh=(yz, U), after_enqueue=self._finalize_early(niter, ref), need_nsq=False,
                             latent_index=mine)
        std_errs, boot_ratios, (LVcorr,) = self._finish_items(res, [LVc], niter, ref)
        z = norm.ppf(1 - (1 - CI) / 2)
        half = (spread.std() if nranks == 1 else np.std(LVcorr, axis=0)) * z    # :723-724
        conf_int = (lvcorrs_orig - half, lvcorrs_orig + half)                  # :725
        debug = {"left_sv_sampled": LVcorr, "right_sv_sampled": None, "indices": inds}
        return conf_int, std_errs, boot_ratios, LVcorr, debug

    def _boot_mb(self, U, s, V, niter, lvcorrs_orig, Tvsc_orig, CI):
        """bootstrap_permutation.py:467-766 for mb and cmb (cmb: projection on the
        normalised contrasts, Tdistrib from X itself, ratios V / std_errs; :658-675, :703)."""
        eng = self._engine
        co = self._cond_order
        bscan = list(self._bscan)
        Yb = np.asarray(self._Ybscan, dtype=float)
        U = np.asarray(U, dtype=float) if self._C is None else self._C
        V = None if V is None else np.asarray(V, dtype=float)
        ng, nc = co.shape
        n = int(co.sum())
        nb, b = Yb.shape
        nbs = len(bscan)
        per = nc + nbs * b
        kr = ng * per                                  # rows of the multiblock
        k = U.shape[1]                                 # latent variables
        bounds_b = cf.cell_bounds(co[:, bscan])
        brows = np.flatnonzero(cf.bscan_mask(co, bscan))                       # bscan index -> row of X
        ref_ready = self._observed_vs(V, s) if self._C is None else self._upload_in_background(lambda: V)
        draws = self._draw_on_rank0(lambda: self._draw_boot_with_guard(niter, Yb, True))
        ti, bi = draws[:, :n], draws[:, n:]
        rank, nranks = dist.world()
        lo, hi = dist.shard_bounds(niter, rank, nranks)
        ref = ref_ready()
        # item matrix = [X[ti] (raw) ; X[brows[bi]] z-scored within the bscan cells]
        src = np.concatenate((ti, brows[bi]), axis=1)[lo:hi]
        cell_lo = np.concatenate(([0, n], n + bounds_b[1:]))
        cell_z = np.ones(len(cell_lo) - 1, dtype=np.int32)
        cell_z[0] = 0
        W = self._W

        def raw_rows(a, z):
            """un-normalised multiblock rows (class_functions.py:479-511) as operators on
            the item matrix: task rows W on the gathered rows, behaviour rows Yz_b."""
            Yz = cf.zscore_cells(Yb[bi[lo:hi][a:z]], bounds_b)
            A = np.zeros((z - a, kr, n + nb))
            for g in range(ng):
                A[:, g * per:g * per + nc, :n] = W[g * nc:(g + 1) * nc]
                for ci in range(nbs):
                    l, h = bounds_b[g * nbs + ci], bounds_b[g * nbs + ci + 1]
                    r0 = g * per + nc + ci * b
                    A[:, r0:r0 + b, n + l:n + h] = np.transpose(Yz[:, l:h], (0, 2, 1))
            return A

        def ops_fn(a, z, rownorm):
            # rows normalised over all voxels (:503-505) then projected on U (:620):
            # op_b = U^T D_b^-1 A_b
            A = raw_rows(a, z)
            with np.errstate(divide="ignore", invalid="ignore"):
                scaled = U[None] / rownorm[:, :, None]
            return np.swapaxes(scaled, 1, 2) @ A                      # (r,j,k)(r,k,i) -> r,j,i

        # the same items as cells of gathered rows (engine.split_rows): behaviour cells first (group x bscan
        # condition: rows brows[bi] of X with rows bi of Ybscan), then the task cells (group x condition: rows ti),
        # whose sums give the task rows -- the mean-centring operator is constant inside a cell
        layout = self._mb_cell_layout() if self._C is None else None
        cells_ok = layout is not None

        def cells_fn(a, z):
            if not cells_ok:
                return None, None
            tb, bb = ti[lo:hi][a:z], bi[lo:hi][a:z]
            return dict(layout, xsrc=np.concatenate((brows[bb], tb), axis=1),
                        ysrc=np.concatenate((bb, np.zeros_like(tb)), axis=1)), Yb

        cnt = hi - lo
        LVc = np.empty((cnt, (len(bounds_b) - 1) * b, k))
        Td = np.empty((cnt, co.size, k))
        bi_m, ti_m = bi[lo:hi], ti[lo:hi]
        spread, spread_t = self._RunningStd(), self._RunningStd()

        def on_batch(a, z, zt, nsq, form):
            Zn = self._normalised_latents(zt, nsq)
            if form == "index+own":
                # the engine computed exactly what is read: the scores of the behaviour sample's rows (:647-650) and,
                # in the last columns, the raw task rows of the sample times V_hat -- the cell means of
                # smeanmat(X_new_T) @ V_hat (:654-656), the mean-centring being linear
                LVc[a:z] = cf.corr_rows(Zn[:, :nb], cf.zscore_cells(Yb[bi_m[a:z]], bounds_b), bounds_b)
                Td[a:z] = Zn[:, nb:]
                spread.add(LVc[a:z])
                spread_t.add(Td[a:z])
                return
            # behaviour latents (:647-650): Xbscan_new @ V_hat = (X @ V_hat)[brows[bi]]
            Lb = np.take_along_axis(Zn, brows[bi_m[a:z]][:, :, None].astype(np.int64), axis=1)
            LVc[a:z] = cf.corr_rows(Lb, cf.zscore_cells(Yb[bi_m[a:z]], bounds_b), bounds_b)
            if self._C is None:
                # task distribution (:654-656): cell means of smeanmat(X_new_T) @ V_hat
                Lt = np.take_along_axis(Zn, ti_m[a:z][:, :, None].astype(np.int64), axis=1)
                Td[a:z] = cf.cell_means_rows(cf.smeanmat_rows(Lt, co, self._mctype), co)
            else:
                # cmb (:665-666): cell means of X @ normalize(crossblock.T), X itself
                Td[a:z] = cf.cell_means_rows(Zn, co)
            spread.add(LVc[a:z])
            spread_t.add(Td[a:z])

        res = eng.boot_items(src, cell_lo, cell_z, k, ops_fn, ref=ref, raw_rows_fn=raw_rows, latent_rows=n,
                             on_batch=on_batch, project_on=U, after_enqueue=self._finalize_early(niter, ref),
                             cells_fn=cells_fn, latent_index=brows[bi_m] if cells_ok else None, own_rows=cells_ok)
        std_errs, boot_ratios, (LVcorr, Tdistrib) = self._finish_items(res, [LVc, Td], niter, ref)
        z = norm.ppf(1 - (1 - CI) / 2)
        half = (spread.std() if nranks == 1 else np.std(LVcorr, axis=0)) * z
        conf_int = (lvcorrs_orig - half, lvcorrs_orig + half)                  # :723-725
        half_t = (spread_t.std() if nranks == 1 else np.std(Tdistrib, axis=0)) * z
        conf_int_T = (Tvsc_orig - half_t, Tvsc_orig + half_t)                  # :732-734
        debug = {"left_sv_sampled": LVcorr, "right_sv_sampled": None, "indices": draws, "Tdistrib": Tdistrib}
        return conf_int, conf_int_T, std_errs, boot_ratios, LVcorr, debug

    def __repr__(self):
        stg = "Permutation Test Results\n------------------------\n\n"
        stg += f"Ratio: {self.permute_ratio}\n\nStep Down Ratio: {self.stepdown_ratio}\n\n"
        stg += "Bootstrap Test Results\n----------------------\n\n"
        stg += f"Selected Confidence Interval Level: {self.CI}\n"
        stg += f"\nLower CI: \n{self.conf_ints[0]}\n\nUpper CI: \n{self.conf_ints[1]}"
        if hasattr(self, "conf_ints_T"):
            stg += f"\n\nLower CI (Task): \n{self.conf_ints_T[0]}\n\nUpper CI (Task): \n{self.conf_ints_T[1]}"
        stg += f"\n\nStandard Errors:\n{self.std_errs}\n\nBootstrap Ratios:\n{self.boot_ratios}"
        return stg

    __str__ = __repr__


ResampleTest._IMPLEMENTATIONS.update({name: _ResampleTestPLS for name in METHOD_NAMES})
