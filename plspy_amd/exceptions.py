"""Exception types of the PLS surface (same names and meaning as the reference's
plspy/core/exceptions.py:4-45, so callers' ``except`` clauses keep working)."""


class Error(Exception):
    """Base class of the package's exceptions."""


class InputMatrixDimensionMismatchError(Error):
    """Row counts of X / Y / the condition table do not agree."""


class ImproperShapeError(Error):
    """A matrix is not 2-dimensional."""


class ConditionMatrixMalformedError(Error):
    """The condition table is not of the expected shape."""


class NotImplementedError(Error):  # noqa: A001  (name mirrors the reference)
    """The requested PLS variant is not available."""


class MissingParameterError(Error):
    """A required argument (e.g. Y for behaviour PLS) is missing."""


class OutOfRangeError(Error):
    """An index is outside the valid range."""
