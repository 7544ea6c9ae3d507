"""The oracle's restatement of plspy/io/io.py:353-398, :427-460, :680-760 on the CPU, against the
reference's own round-trip property (plspy/tests/test_io.py:8-36).  The reference module imports
nibabel and cannot be imported here: these functions are not pinned by generated fixtures."""
import numpy as np

from oracle import plspy_oracle as orc


def test_round_trip_property_of_the_reference():
    rand_s = np.random.RandomState(950613)
    mock_subjects = [rand_s.rand(20, 10, 10, 10) for _ in range(5)]
    mask = orc.io_create_threshold_mask_from_matrices(mock_subjects, threshold=0.15)
    assert mask.shape == (10, 10, 10) and mask.dtype == bool and 0 < mask.sum() < mask.size
    masked = orc.io_apply_mask_matrices(mock_subjects, mask)
    assert all(m.shape == (20 * mask.sum(),) for m in masked)
    recovered = orc.io_remap_vectorized_subject_to_4d(masked[0], mask, mock_subjects[0].shape)
    # the reference's element-wise check (test_io.py:21-36), vectorised
    assert np.allclose(recovered[:, mask], mock_subjects[0][:, mask])
    assert (recovered[:, ~mask] == 0).all()


def test_concat_flatten_all_groups():
    rs = np.random.RandomState(1)
    groups = [rs.randn(3, 4, 5), rs.randn(2, 4, 5)]
    X = orc.io_concat_flatten_all_groups(groups)
    assert X.shape == (5, 20)
    np.testing.assert_array_equal(X[3], groups[1][0].reshape(-1))
