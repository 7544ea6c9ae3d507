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
