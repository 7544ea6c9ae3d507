"""F4: the upstream feed on the device -- stream compaction of the mask and the
row gather that builds X -- against the NumPy restatement of plspy/io/io.py:427-460, :680-698 and the
reference's own round-trip property (plspy/tests/test_io.py:8-36).  The reference module itself cannot
be imported here (nibabel): parity with it is unpinned, see plspy_amd/io.py."""
import numpy as np
import pytest

from oracle import plspy_oracle as orc

pytestmark = pytest.mark.gpu


def test_round_trip_of_the_reference_io_test():
    """plspy/tests/test_io.py:8-36 with the device path in place of apply_mask_matrices."""
    from plspy_amd import io
    rand_s = np.random.RandomState(950613)
    mock_subjects = [rand_s.rand(20, 10, 10, 10) for _ in range(5)]
    mask = orc.io_create_threshold_mask_from_matrices(mock_subjects, threshold=0.15)
    masked = io.apply_mask_matrices(mock_subjects, mask)
    want = orc.io_apply_mask_matrices(mock_subjects, mask)
    for got, w in zip(masked, want):
        np.testing.assert_array_equal(got.cpu().numpy(), w)
    recovered = orc.io_remap_vectorized_subject_to_4d(masked[0].cpu().numpy(), mask, mock_subjects[0].shape)
    assert np.array_equal(recovered[:, mask], mock_subjects[0][:, mask])
    assert not recovered[:, ~mask].any()


@pytest.mark.parametrize("shape,T,dtype", [((7, 5, 3), 4, np.float64), ((64, 64, 30), 3, np.float32),
                                           ((1, 1, 1), 2, np.float64), ((41, 17), 1, np.float64),
                                           ((2049,), 5, np.float64)])
def test_mask_compaction_and_design_matrix(shape, T, dtype):
    """Indices of the compaction are NumPy's flatnonzero (empty, full and ragged masks, sizes on both
    sides of a workgroup's 2048 mask bytes); X equals apply_mask_matrices + concat_flatten_all_groups
    bit for bit (fp32 sources widen exactly)."""
    from plspy_amd import io
    rs = np.random.RandomState(len(shape) * 100 + T)
    for density in (0.0, 0.3, 1.0):
        mask = rs.rand(*shape) < density
        subjects = [rs.randn(T, *shape).astype(dtype) for _ in range(3)]
        idx = io.mask_indices(mask).cpu().numpy()
        np.testing.assert_array_equal(idx, np.flatnonzero(mask))
        X = io.masked_design_matrix(subjects, mask).cpu().numpy()
        want = orc.io_concat_flatten_all_groups([np.stack([v.astype(np.float64)]) for v in
                                                 orc.io_apply_mask_matrices(subjects, mask)])
        np.testing.assert_array_equal(X, want)
        flat = io.concat_flatten_all_groups([m.reshape(1, T, -1) for m in io.apply_mask_matrices(subjects, mask)])
        np.testing.assert_array_equal(flat.cpu().numpy(), want)


def test_design_matrix_feeds_the_engine():
    """X from the feed goes into PLS() without a host round trip."""
    import plspy_amd
    from plspy_amd import io
    rs = np.random.RandomState(3)
    subjects = [rs.randn(2, 6, 5, 4) for _ in range(12)]
    mask = rs.rand(6, 5, 4) < 0.6
    X = io.masked_design_matrix(subjects, mask)
    np.random.seed(5)
    res = plspy_amd.PLS(X, [3, 3], 2, num_perm=5, num_boot=5, pls_method="mct")
    np.random.seed(5)
    ref = plspy_amd.PLS(X.cpu().numpy(), [3, 3], 2, num_perm=5, num_boot=5, pls_method="mct")
    np.testing.assert_allclose(res.s, ref.s, rtol=1e-12)


@pytest.mark.parametrize("mshape", [(1, 5, 6, 7), (6, 7), (7,), (4, 5, 6, 7), (4, 1, 6, 1)])
def test_masks_that_broadcast(mshape):
    """Masks NumPy broadcasts against (T, X, Y, Z) volumes in other ways than the plain spatial one -- a
    leading axis of one, trailing axes only, the full shape, singleton axes inside -- against the oracle's
    restatement ``m[np.broadcast_to(mask, m.shape)]`` (io.py:456-458); and one that does not broadcast."""
    from plspy_amd import io as pio
    rs = np.random.RandomState(len(mshape) + sum(mshape))
    vols = [rs.randn(4, 5, 6, 7), rs.randn(4, 5, 6, 7).astype(np.float32)]
    mask = rs.rand(*mshape) > 0.4
    got = pio.apply_mask_matrices(vols, mask)
    want = orc.io_apply_mask_matrices(vols, mask)
    for g, w in zip(got, want):
        np.testing.assert_array_equal(g.cpu().numpy(), w.astype(np.float64))
    with pytest.raises(ValueError):
        pio.apply_mask_matrices(vols, rs.rand(3, 6, 7) > 0.5)
