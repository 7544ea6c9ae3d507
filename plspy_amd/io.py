"""The upstream feed: the two pure-array steps of ``plspy/io/io.py``
that turn subjects' volumes into the design matrix, with X produced on the device.

  apply_mask_matrices(matrices, mask)         io.py:427-460
  concat_flatten_all_groups(groups_list)      io.py:680-698
  masked_design_matrix(subjects, mask)        both at once, straight into the rows of X

NIfTI loading itself (nibabel) stays out of scope.  The reference's module cannot be imported here
(it imports nibabel at its top), so no fixture could be generated from it: parity of these functions
is UNPINNED against the reference and rests on the NumPy restatement in oracle/plspy_oracle.py
(``io_apply_mask_matrices`` etc., which cites the lines it follows) and on the reference's own
round-trip property (plspy/tests/test_io.py:8-36), see tests/test_gpu_io.py.

Results are fp64 device tensors (the engine's type); the reference keeps the input dtype."""
import ctypes

import numpy as np
import torch

from . import _lib


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _device(device):
    if not torch.cuda.is_available():
        raise RuntimeError("plspy_amd needs a ROCm GPU (MI355X); no CPU fallback exists")
    return torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")


def mask_indices(mask, device=None):
    """Flat C-order indices of the voxels `mask` selects (ascending), as a device int64 tensor: the
    stream compaction of the mask (plsr_mask_indices)."""
    lib = _lib.load()
    dev = _device(device)
    m = torch.as_tensor(np.ascontiguousarray(np.asarray(mask) != 0).view(np.uint8).reshape(-1)).to(dev)
    nvox = int(m.numel())
    idx = torch.empty(nvox, dtype=torch.int64, device=dev)
    count = torch.zeros(1, dtype=torch.int64, device=dev)
    need = lib.plsr_mask_indices_workspace_bytes(nvox)
    work = torch.empty(max(need, 8), dtype=torch.uint8, device=dev)
    _lib.check(lib.plsr_mask_indices(_ptr(m), nvox, _ptr(idx), _ptr(count), _ptr(work), need, _stream()),
               "plsr_mask_indices")
    return idx[:int(count.item())]


def _apply(m, idx, out):
    """out (T, nsel) rows <- volume m (T, *spatial) gathered at idx."""
    lib = _lib.load()
    m = np.asarray(m)
    if m.dtype not in (np.float32, np.float64):
        m = m.astype(np.float64)
    T = m.shape[0]
    d_m = torch.as_tensor(np.ascontiguousarray(m).reshape(T, -1)).to(out.device)
    _lib.check(lib.plsr_mask_apply_rows(_ptr(d_m), int(m.dtype == np.float32), d_m.stride(0), T, _ptr(idx),
                                        idx.numel(), _ptr(out), out.stride(0), _stream()), "plsr_mask_apply_rows")


def apply_mask_matrices(matrices, mask, device=None):
    """io.py:427-460: for every matrix m, ``m[np.broadcast_to(mask, m.shape)]`` -- the selected elements in C
    order, flattened.  Any mask NumPy broadcasts to m's shape is taken (a spatial mask against (T, X, Y, Z)
    volumes, a (1, X, Y, Z) or (Y, Z) mask, a full-shape mask); one that does not broadcast raises the
    reference's ValueError.  Where the broadcast mask is the same for every step of the leading (time) axis
    the spatial selection is compacted once and gathered per time point; otherwise the whole array is one
    selection.  Returns a list of 1-D fp64 device tensors."""
    mk = np.asarray(mask) != 0
    cache = {}
    masked = []
    for m in matrices:
        m = np.asarray(m)
        bm = np.broadcast_to(mk, m.shape)                       # raises the reference's ValueError
        per_time = m.ndim >= 2 and (mk.ndim < m.ndim or mk.shape[0] == 1)
        key = (m.shape, per_time)
        if key not in cache:
            cache[key] = mask_indices(bm[0] if per_time else bm, device)
        idx = cache[key]
        vol = m if per_time else m[None]
        out = torch.empty((vol.shape[0], idx.numel()), dtype=torch.float64, device=idx.device)
        _apply(vol, idx, out)
        masked.append(out.reshape(-1))
    return masked


def concat_flatten_all_groups(groups_list):
    """io.py:680-698: all groups stacked along the first axis, every subject flattened to a row."""
    full = torch.cat([torch.as_tensor(g) for g in groups_list], dim=0)
    return full.reshape(full.shape[0], -1)


def masked_design_matrix(subjects, mask, device=None):
    """X (len(subjects) x T * nsel, fp64) on the device from the subjects' volumes (each (T, *spatial),
    host arrays) and the mask: apply_mask_matrices + concat_flatten_all_groups without the masked
    copies -- every subject is gathered straight into its row of X."""
    idx = mask_indices(mask, device)
    T = int(np.asarray(subjects[0]).shape[0])
    nsel = int(idx.numel())
    X = torch.empty((len(subjects), T * nsel), dtype=torch.float64, device=idx.device)
    for i, m in enumerate(subjects):
        if np.asarray(m).shape[0] != T:
            raise ValueError("all subjects must have the same number of time points")
        _apply(m, idx, X[i].view(T, nsel))
    return X
