"""PLS variant classes: the host side of the drop-in.

Each class keeps the reference's constructor signature, validation, attribute
names and the final U/V swap (plspy/core/pls_classes.py), so a result object
can be consumed by code written against plspy (e.g. its visualize package).
The observed decomposition is done once on the host exactly as the reference
does it; the permutation / bootstrap / split-half loops run on the GPU through
bootstrap_permutation.py and split_half_resampling.py."""
import abc

import numpy as np

from . import bootstrap_permutation, exceptions, operators
from .engine import ProjectionEngine


class PLSBase(abc.ABC):
    """Registry / factory (pls_classes.py:12-71)."""

    _subclasses = {}
    _pls_types = {
        "mct": "Mean-Centring Task PLS",
        "rb": "Regular Behaviour PLS",
        "cst": "Contrast Task PLS",
        "csb": "Contrast Behaviour PLS",
        "mb": "Multiblock PLS",
        "cmb": "Contrast Multiblock PLS",
    }

    @classmethod
    def _register_subclass(cls, pls_method):
        def decorator(subclass):
            cls._subclasses[pls_method] = subclass
            return subclass
        return decorator

    @classmethod
    def _create(cls, pls_method, *args, **kwargs):
        if pls_method not in cls._subclasses and pls_method in cls._pls_types:
            raise exceptions.NotImplementedError(
                f"Specified PLS/Resample method {cls._pls_types[pls_method]} "
                "has not yet been implemented.")
        elif pls_method not in cls._subclasses:
            raise ValueError(f"Invalid PLS method {pls_method}")
        return cls._subclasses[pls_method](*args, **kwargs)

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


@PLSBase._register_subclass("mct")
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

        engine = ProjectionEngine(X)
        co = np.asarray(self.cond_order)
        # observed decomposition (pls_classes.py:258-266), on the device: the
        # two k x p blocks come from the projection kernel, the thin SVD
        # (class_functions.py:122) from the Gram + Jacobi + back-projection
        # kernels (engine.thin_svd)
        Wm = operators.cell_mean_operator(co)
        W = operators.mean_centre_operator(co, self.mctype)
        blocks = engine.apply_operator(np.vstack((Wm, W))).cpu().numpy()
        self.X_means, self.X_mc = blocks[:len(Wm)], blocks[len(Wm):]
        self.U, self.s, self.V = engine.thin_svd(W)
        self.X_latent = np.dot(self.X, self.V)
        Tvsc_orig = Wm @ self.X_latent

        self.resample_tests = bootstrap_permutation.ResampleTest._create(
            self.pls_alg, self.X, None, self.U, self.s, self.V, self.cond_order, self.mctype,
            preprocess=None, nperm=self.num_perm, nboot=self.num_boot, Tvsc_orig=Tvsc_orig,
            CI=self.CI, engine=engine)

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
