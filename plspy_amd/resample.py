"""Resample index generation (host side).

The reference draws its indices from NumPy's *global legacy* RandomState, one
resample at a time inside the Python loops (plspy/core/resample.py:9-165,
split_half_resampling.py:130-169, :271-283).  To be a drop-in, the same seed
must give the same resamples, so a whole phase's index table (R x n int32,
uploaded to the GPU in one copy) is drawn from that same global stream in the
same order.  X itself is never gathered on the host.

Two implementations of the same draws:
  * ``native=True`` (default): the C restatement of NumPy's legacy MT19937
    permutation / choice in the C-ABI library (plsr_rng_*): reads
    ``np.random.get_state()``, advances it, writes it back with
    ``np.random.set_state()`` -- bit-identical output and stream position,
    without the per-resample Python loop (which was 5x the GPU time);
  * ``native=False``: the same calls issued through ``np.random`` itself (kept
    for the parity tests that pin the native generator)."""
import ctypes

import numpy as np

NATIVE = True


def subject_tables(cond_order):
    """Per group, the (subjects x conditions) table of row ids."""
    tables = []
    start = 0
    for sizes in np.asarray(cond_order):
        cols = []
        for sz in sizes:
            cols.append(np.arange(start, start + int(sz)))
            start += int(sz)
        tables.append(np.column_stack(cols))
    return tables


# ---------------------------------------------------------------------------
# through np.random (reference call order)
# ---------------------------------------------------------------------------
def draw_task_permutation(table):
    """One task-PLS permutation (resample.py:63-73): shuffle each subject's
    conditions, then shuffle every condition slot across all subjects; the
    result is flattened slot-major and used as a row order verbatim."""
    nsub, nc = table.shape
    within = np.empty_like(table)
    for r in range(nsub):
        within[r] = np.random.permutation(table[r])
    out = np.empty((nc, nsub), dtype=table.dtype)
    for c in range(nc):
        out[c] = np.random.permutation(within[:, c])
    return out.ravel()


def draw_bootstrap(tables):
    """One bootstrap sample (resample.py:132-160): per group, subjects with
    replacement, the same draw for every condition."""
    parts = []
    for tbl in tables:
        ns = tbl.shape[0]
        pick = np.random.choice(ns, ns, replace=True)
        parts.append(tbl[pick].T.ravel())
    return np.concatenate(parts)


# ---------------------------------------------------------------------------
# native generator on the same global stream
# ---------------------------------------------------------------------------
class _GlobalStream:
    """np.random's legacy MT19937 state, lent to the native generator."""

    def __enter__(self):
        st = np.random.get_state()
        if st[0] != "MT19937":
            raise RuntimeError("np.random's global state is not the legacy MT19937")
        self._rest = st[3:]
        self.key = np.ascontiguousarray(st[1], dtype=np.uint32).copy()
        self.pos = ctypes.c_int32(int(st[2]))
        return self

    def __exit__(self, exc_type, exc, tb):
        if exc_type is None:
            np.random.set_state(("MT19937", self.key, int(self.pos.value)) + tuple(self._rest))
        return False

    def args(self):
        return self.key.ctypes.data_as(ctypes.c_void_p), ctypes.byref(self.pos)


def _lib():
    from . import _lib as L
    return L.load(), L


def permutations(n, count, native=None):
    """``count`` draws of np.random.permutation(n): (count, n) int32."""
    out = np.empty((count, n), dtype=np.int32)
    if count == 0:
        return out
    if NATIVE if native is None else native:
        lib, L = _lib()
        with _GlobalStream() as gs:
            L.check(lib.plsr_rng_permutation(*gs.args(), n, count, out.ctypes.data_as(ctypes.c_void_p)),
                    "plsr_rng_permutation")
    else:
        for i in range(count):
            out[i] = np.random.permutation(n)
    return out


def task_permutations(cond_order, count, native=None):
    table = np.ascontiguousarray(np.concatenate(subject_tables(cond_order)), dtype=np.int32)
    out = np.empty((count, table.size), dtype=np.int32)
    if count == 0:
        return out
    if NATIVE if native is None else native:
        lib, L = _lib()
        with _GlobalStream() as gs:
            L.check(lib.plsr_rng_task_permutations(*gs.args(), table.ctypes.data_as(ctypes.c_void_p),
                                                   table.shape[0], table.shape[1], count,
                                                   out.ctypes.data_as(ctypes.c_void_p)),
                    "plsr_rng_task_permutations")
    else:
        for i in range(count):
            out[i] = draw_task_permutation(table)
    return out


def bootstraps(cond_order, count, native=None):
    tables = subject_tables(cond_order)
    n = sum(t.size for t in tables)
    out = np.empty((count, n), dtype=np.int32)
    if count == 0:
        return out
    ncs = {t.shape[1] for t in tables}
    if (NATIVE if native is None else native) and len(ncs) == 1:
        lib, L = _lib()
        table = np.ascontiguousarray(np.concatenate(tables), dtype=np.int32)
        groups = np.array([t.shape[0] for t in tables], dtype=np.int32)
        with _GlobalStream() as gs:
            L.check(lib.plsr_rng_bootstraps(*gs.args(), table.ctypes.data_as(ctypes.c_void_p),
                                            groups.ctypes.data_as(ctypes.c_void_p), len(groups),
                                            table.shape[1], count, out.ctypes.data_as(ctypes.c_void_p)),
                    "plsr_rng_bootstraps")
    else:
        for i in range(count):
            out[i] = draw_bootstrap(tables)
    return out
