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


def permutation_rounds(sizes, count, native=None):
    """``count`` rounds of np.random.permutation(sizes[0]), ..., np.random.permutation(sizes[-1]) in that order:
    a list of (count, size) int32 arrays, one per entry of sizes."""
    sizes = [int(x) for x in sizes]
    out = np.empty((count, sum(sizes)), dtype=np.int32)
    if count:
        if NATIVE if native is None else native:
            lib, L = _lib()
            arr = (ctypes.c_int32 * len(sizes))(*sizes)
            with _GlobalStream() as gs:
                L.check(lib.plsr_rng_permutation_seq(*gs.args(), arr, len(sizes), count,
                                                     out.ctypes.data_as(ctypes.c_void_p)), "plsr_rng_permutation_seq")
        else:
            for i in range(count):
                out[i] = np.concatenate([np.random.permutation(s) for s in sizes])
    offs = np.concatenate(([0], np.cumsum(sizes)))
    return [out[:, a:b] for a, b in zip(offs[:-1], offs[1:])]


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


def mb_permutation_tries(cond_order, nrows, count, native=None):
    """``count`` tries of the multiblock permutation (bootstrap_permutation.py:343-347):
    per try a task permutation, then np.random.permutation(nrows).
    Returns (task (count, n), rows (count, nrows))."""
    table = np.ascontiguousarray(np.concatenate(subject_tables(cond_order)), dtype=np.int32)
    task = np.empty((count, table.size), dtype=np.int32)
    rows = np.empty((count, nrows), dtype=np.int32)
    if count == 0:
        return task, rows
    if NATIVE if native is None else native:
        lib, L = _lib()
        with _GlobalStream() as gs:
            L.check(lib.plsr_rng_mb_permutations(*gs.args(), table.ctypes.data_as(ctypes.c_void_p),
                                                 table.shape[0], table.shape[1], nrows, count,
                                                 task.ctypes.data_as(ctypes.c_void_p),
                                                 rows.ctypes.data_as(ctypes.c_void_p)),
                    "plsr_rng_mb_permutations")
    else:
        for i in range(count):
            task[i] = draw_task_permutation(table)
            rows[i] = np.random.permutation(nrows)
    return task, rows


def mb_bootstrap_tries(cond_order, bscan, count, native=None):
    """``count`` tries of the multiblock bootstrap (bootstrap_permutation.py:547-553):
    per try a task bootstrap, then a behaviour bootstrap on the bscan conditions.
    Returns (task (count, n), beh (count, n_bscan))."""
    co = np.asarray(cond_order)
    tables = subject_tables(co)
    btables = subject_tables(co[:, list(bscan)])
    n, nb = sum(t.size for t in tables), sum(t.size for t in btables)
    task = np.empty((count, n), dtype=np.int32)
    beh = np.empty((count, nb), dtype=np.int32)
    if count == 0:
        return task, beh
    ncs = {t.shape[1] for t in tables}
    if (NATIVE if native is None else native) and len(ncs) == 1:
        lib, L = _lib()
        table = np.ascontiguousarray(np.concatenate(tables), dtype=np.int32)
        btable = np.ascontiguousarray(np.concatenate(btables), dtype=np.int32)
        groups = np.array([t.shape[0] for t in tables], dtype=np.int32)
        with _GlobalStream() as gs:
            L.check(lib.plsr_rng_mb_bootstraps(*gs.args(), table.ctypes.data_as(ctypes.c_void_p),
                                               groups.ctypes.data_as(ctypes.c_void_p), len(groups),
                                               table.shape[1], btable.ctypes.data_as(ctypes.c_void_p),
                                               btable.shape[1], count, task.ctypes.data_as(ctypes.c_void_p),
                                               beh.ctypes.data_as(ctypes.c_void_p)),
                    "plsr_rng_mb_bootstraps")
    else:
        for i in range(count):
            task[i] = draw_bootstrap(tables)
            beh[i] = draw_bootstrap(btables)
    return task, beh


def draw_guarded(niter, draw_tries, is_bad, max_tries=100):
    """``niter`` accepted tries from a sequential candidate stream, as the
    reference's ``for attempt in range(100): draw; if ok: break`` loops do
    (bootstrap_permutation.py:333-355, :542-572), in batches: a batch draws
    exactly as many candidates as are still missing, so the RNG stream is
    consumed exactly as by the one-at-a-time loop.  draw_tries(m) -> tuple of
    (m, .) arrays; is_bad(*arrays) -> (m,) bool.  Returns the tuple of accepted
    arrays, or None after ``max_tries`` consecutive bad candidates."""
    kept = None
    have, run = 0, 0
    while have < niter:
        tries = draw_tries(niter - have)
        bad = np.asarray(is_bad(*tries), dtype=bool)
        for b in bad:                                  # consecutive failures across batches
            run = run + 1 if b else 0
            if run >= max_tries:
                return None
        good = np.flatnonzero(~bad)
        if kept is None:
            kept = [np.empty((niter,) + t.shape[1:], dtype=t.dtype) for t in tries]
        for k, t in zip(kept, tries):
            k[have:have + len(good)] = t[good]
        have += len(good)
    return tuple(kept)
