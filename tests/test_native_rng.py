"""The native (C) restatement of NumPy's legacy RandomState draws must be
bit-identical to np.random itself -- values AND stream position -- for the
resample shapes the reference produces.  CPU only."""
import numpy as np
import pytest

from plspy_amd import _build, resample


@pytest.fixture(scope="module", autouse=True)
def lib():
    _build.build()


def _both(fn, *args):
    """Run fn natively and through np.random from the same state; return the
    two outputs and the two end states."""
    st = np.random.get_state()
    a = fn(*args, native=True)
    sa = np.random.get_state()
    np.random.set_state(st)
    b = fn(*args, native=False)
    sb = np.random.get_state()
    return a, b, sa, sb


def _same_state(sa, sb):
    assert sa[0] == sb[0] and sa[2] == sb[2] and sa[3] == sb[3] and sa[4] == sb[4]
    np.testing.assert_array_equal(sa[1], sb[1])


@pytest.mark.parametrize("seed", [0, 1234, 2**31 - 1])
@pytest.mark.parametrize("groups,nc", [((10, 10), 3), ((3, 2), 2), ((8,), 3), ((20, 20, 20, 20), 3),
                                        ((1, 5), 4), ((7,), 1), ((2, 1), 1)])
def test_task_permutations_and_bootstraps(seed, groups, nc):
    co = np.array([[g] * nc for g in groups])
    np.random.seed(seed)
    np.random.standard_normal(3)          # leave a cached gaussian in the state
    a, b, sa, sb = _both(resample.task_permutations, co, 57)
    np.testing.assert_array_equal(a, b)
    _same_state(sa, sb)
    a, b, sa, sb = _both(resample.bootstraps, co, 61)
    np.testing.assert_array_equal(a, b)
    _same_state(sa, sb)


@pytest.mark.parametrize("n", [1, 2, 3, 17, 60, 240, 257, 100000])
def test_permutation_rows(n):
    np.random.seed(n)
    a, b, sa, sb = _both(resample.permutations, n, 9)
    np.testing.assert_array_equal(a, b)
    _same_state(sa, sb)


def test_many_draws_cross_state_refills():
    """Enough draws to regenerate the 624-word key hundreds of times."""
    co = np.array([[10] * 3, [10] * 3])
    np.random.seed(42)
    a, b, sa, sb = _both(resample.task_permutations, co, 3000)
    np.testing.assert_array_equal(a, b)
    _same_state(sa, sb)
    after_native = np.random.random()
    np.random.set_state(sb)
    assert after_native == np.random.random()


def test_native_is_default_and_fast():
    import time
    co = np.array([[10] * 3, [10] * 3])
    np.random.seed(1)
    t0 = time.perf_counter()
    resample.task_permutations(co, 1000)
    resample.bootstraps(co, 1000)
    t_native = time.perf_counter() - t0
    np.random.seed(1)
    t0 = time.perf_counter()
    resample.task_permutations(co, 1000, native=False)
    resample.bootstraps(co, 1000, native=False)
    t_numpy = time.perf_counter() - t0
    assert t_native < t_numpy / 3, (t_native, t_numpy)


def test_multiblock_tries_match_the_python_loops():
    """plsr_rng_mb_permutations / plsr_rng_mb_bootstraps: two draws per try,
    interleaved on np.random's stream, bit-exact against the Python loops."""
    from plspy_amd import resample
    co = np.array([[5, 5, 5], [4, 4, 4]])
    for seed in (0, 7):
        np.random.seed(seed)
        a_t, a_r = resample.mb_permutation_tries(co, 18, 9, native=False)
        tail_a = np.random.randint(0, 1 << 30, size=3)
        np.random.seed(seed)
        b_t, b_r = resample.mb_permutation_tries(co, 18, 9, native=True)
        tail_b = np.random.randint(0, 1 << 30, size=3)
        np.testing.assert_array_equal(a_t, b_t)
        np.testing.assert_array_equal(a_r, b_r)
        np.testing.assert_array_equal(tail_a, tail_b)          # stream left in the same state
        np.random.seed(seed)
        a_t, a_b = resample.mb_bootstrap_tries(co, [0, 2], 11, native=False)
        tail_a = np.random.randint(0, 1 << 30, size=3)
        np.random.seed(seed)
        b_t, b_b = resample.mb_bootstrap_tries(co, [0, 2], 11, native=True)
        tail_b = np.random.randint(0, 1 << 30, size=3)
        np.testing.assert_array_equal(a_t, b_t)
        np.testing.assert_array_equal(a_b, b_b)
        np.testing.assert_array_equal(tail_a, tail_b)


def test_guarded_batches_consume_the_stream_like_the_loop():
    """draw_guarded: candidates are consumed in order, bad ones skipped, no
    over-draw -- identical results and identical stream state to the reference's
    one-at-a-time redraw loop, including when many candidates are rejected."""
    from plspy_amd import resample

    def bad(rows):                       # rejects about 40 % of the candidates
        return (rows[:, 0] % 5) < 2

    for seed in (1, 2, 3):
        np.random.seed(seed)
        want = np.empty((40, 12), dtype=np.int32)
        for i in range(40):
            while True:
                r = np.random.permutation(12)
                if not bad(r[None])[0]:
                    break
            want[i] = r
        tail_a = np.random.randint(0, 1 << 30, size=3)
        np.random.seed(seed)
        (got,) = resample.draw_guarded(40, lambda m: (resample.permutations(12, m),), bad)
        tail_b = np.random.randint(0, 1 << 30, size=3)
        np.testing.assert_array_equal(got, want)
        np.testing.assert_array_equal(tail_a, tail_b)
    # 100 consecutive rejections -> None (the caller raises the reference's exception)
    np.random.seed(0)
    assert resample.draw_guarded(3, lambda m: (resample.permutations(4, m),),
                                 lambda rows: np.ones(len(rows), dtype=bool)) is None


def test_permutation_rounds_match_numpy_and_stream_position():
    """plsr_rng_permutation_seq: rounds of permutations of different lengths (the per-group subject
    shuffles of a split, the subject + row shuffles of a null split) -- same values and same stream
    position as the reference's loop (split_half_resampling.py:136, :271, :282)."""
    from plspy_amd import resample
    for seed, sizes in ((1, [20, 20]), (2, [6, 5, 7]), (3, [40, 120]), (4, [1, 3])):
        np.random.seed(seed)
        want = [np.empty((9, s), dtype=np.int32) for s in sizes]
        for i in range(9):
            for j, s in enumerate(sizes):
                want[j][i] = np.random.permutation(s)
        tail_a = np.random.randint(0, 1 << 30, size=3)
        for native in (True, False):
            np.random.seed(seed)
            got = resample.permutation_rounds(sizes, 9, native=native)
            tail_b = np.random.randint(0, 1 << 30, size=3)
            for g, w in zip(got, want):
                np.testing.assert_array_equal(g, w)
            np.testing.assert_array_equal(tail_a, tail_b)


def test_split_draws_follow_the_reference_order():
    """_draw_splits (vectorised, native shuffles) against the reference's per-split loop
    (split_half_resampling.py:119-169, :266-283, :316) written out with np.random, for task,
    behaviour and multiblock PLS and unequal groups."""
    from plspy_amd import resample
    from plspy_amd import split_half_resampling as sh
    for alg, groups, nc, bscan in (("mct", (6, 5), 3, None), ("rb", (4, 4), 2, None), ("mb", (7, 6, 5), 3, [0, 2])):
        co = np.array([[g] * nc for g in groups])
        n, S = int(co.sum()), 5
        tables = resample.subject_tables(co)
        alltab = np.concatenate(tables)
        np.random.seed(11)
        want = []
        for _ in range(S):
            p1, p2, q1, q2 = [], [], [], []
            for tbl in tables:
                half = tbl.shape[0] // 2
                t = tbl[np.random.permutation(tbl.shape[0])]
                p1.append(t[:half].flatten())
                p2.append(t[half:].flatten())
                if bscan:
                    q1.append(t[:half][:, bscan].flatten())
                    q2.append(t[half:][:, bscan].flatten())
            d = dict(x1=np.concatenate(p1), x2=np.concatenate(p2))
            d.update(y1=d["x1"], y2=d["x2"])
            if bscan:
                d.update(b1=np.concatenate(q1), b2=np.concatenate(q2))
                d.update(xb1=d["b1"], xb2=d["b2"])
            want.append(d)
        half = sum(t.shape[0] // 2 for t in tables)
        for _ in range(S):
            t = alltab[np.random.permutation(n // nc)]
            i1, i2 = t[:half].flatten(), t[half:].flatten()
            d = dict(x1=i1, x2=i2, y1=i1, y2=i2)
            if bscan:
                d.update(b1=t[:half][:, bscan].flatten(), b2=t[half:][:, bscan].flatten())
            perm = np.random.permutation(n)
            if alg == "rb":
                d.update(y1=perm[i1], y2=perm[i2])
            else:
                d.update(x1=perm[i1], x2=perm[i2])
                if bscan:
                    d.update(xb1=perm[d["b1"]], xb2=perm[d["b2"]])
            want.append(d)
        tail_a = np.random.randint(0, 1 << 30, size=3)
        np.random.seed(11)
        got, g1, g2 = sh._draw_splits(alg, co, S, n, bscan)
        tail_b = np.random.randint(0, 1 << 30, size=3)
        np.testing.assert_array_equal(tail_a, tail_b)
        assert g1 == [g // 2 for g in groups] and g2 == [g - g // 2 for g in groups]
        assert len(got) == 2 * S
        for i, d in enumerate(want):
            for key, val in d.items():
                np.testing.assert_array_equal(got.stack(key)[i], val, err_msg=f"{alg} split {i} {key}")
