"""Resample index generation (host side).

The reference draws its indices from NumPy's *global legacy* RandomState, one
resample at a time inside the Python loops (plspy/core/resample.py:9-165,
split_half_resampling.py:130-169, :271-283).  To be a drop-in, the same seed
must give the same resamples, so these functions issue the same
``np.random.permutation`` / ``np.random.choice`` calls in the same order --
but for a whole phase at once, returning an (R, n) int32 table that is
uploaded to the GPU in one copy.  X itself is never gathered on the host."""
import numpy as np


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


def task_permutations(cond_order, count):
    table = np.concatenate(subject_tables(cond_order))
    out = np.empty((count, table.size), dtype=np.int32)
    for i in range(count):
        out[i] = draw_task_permutation(table)
    return out


def bootstraps(cond_order, count):
    tables = subject_tables(cond_order)
    n = sum(t.size for t in tables)
    out = np.empty((count, n), dtype=np.int32)
    for i in range(count):
        out[i] = draw_bootstrap(tables)
    return out
