"""``PLS()`` and ``methods`` -- the public seam, with the reference's argument
handling (plspy/core/pls.py:11-18, :21-93): ``pls_method`` selects the variant
(default "mct"), count arguments are validated here, everything else is passed
to the variant class."""
from . import pls_classes

methods = {
    "mct": pls_classes._MeanCentreTaskPLS,
    "rb": pls_classes._RegularBehaviourPLS,
    "mb": pls_classes._MultiblockPLS,
    "cst": pls_classes._ContrastTaskPLS,
    "csb": pls_classes._ContrastBehaviourPLS,
    "cmb": pls_classes._ContrastMultiblockPLS,
}


def PLS(*args, **kwargs):
    """Run a PLS analysis; returns the variant's result object.

    Same validation and error messages as the reference (pls.py:44-79); note
    that ``CI`` and ``lv`` are only validated when ``num_split`` is given."""
    pls_method = kwargs.pop("pls_method", "mct")
    kwargs["pls_alg"] = pls_method

    def _count(name, message):
        if name in kwargs:
            v = kwargs[name]
            if v < 0 or not isinstance(v, int):
                raise ValueError(message)

    if "num_split" in kwargs:
        _count("num_split", "Invalid number of splits provided. Value must be a positive integer.")
        if "CI" in kwargs:
            ci = kwargs["CI"]
            if ci is None or ci < 0 or ci > 1:
                raise ValueError("CI should be within 0 and 1.")
        if "lv" in kwargs:
            lv = kwargs["lv"]
            if lv <= 0 or not isinstance(lv, int):
                raise ValueError("lv must be a positive integer greater than 0.")
    _count("num_boot", "Invalid number of bootstraps provided. Value must be a positive integer.")
    _count("num_perm", "Invalid number of permutations provided. Value must be a positive integer.")
    return pls_classes.PLSBase._create(pls_method, *args, **kwargs)
