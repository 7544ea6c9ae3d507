"""ctypes binding of include/plsr.h.

The library is loaded from the source tree (plspy_amd/csrc/libplsr_hip.so).
torch is imported first so that the process has exactly one HIP runtime: both
torch's bundled libamdhip64 and /opt/rocm's carry the soname libamdhip64.so.7,
and the dynamic loader reuses the one already mapped.

There is no CPU fallback: a missing library is an ImportError at first use."""
import ctypes
import os

import torch  # noqa: F401  (must precede the CDLL below, see module docstring)

from . import _build

c_i32, c_i64, c_sz, c_vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_size_t, ctypes.c_void_p


class Layout(ctypes.Structure):
    """plsr_layout_t"""
    _fields_ = [("n", c_i32), ("k", c_i32), ("R", c_i32), ("nk", c_i32), ("kp", c_i32),
                ("period", c_i32), ("Rp", c_i32), ("ntiles", c_i32), ("frag_elems", c_i64)]


# every symbol include/plsr.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "plsr_abi_version": (c_i32, []),
    "plsr_strerror": (ctypes.c_char_p, [c_i32]),
    "plsr_last_hip_error": (c_i32, []),
    "plsr_layout_init": (c_i32, [c_i32, c_i32, c_i32, ctypes.POINTER(Layout)]),
    "plsr_ops_from_indices": (c_i32, [c_vp, c_vp, ctypes.POINTER(Layout), c_vp, c_vp]),
    "plsr_ops_pack": (c_i32, [c_vp, ctypes.POINTER(Layout), c_vp, c_vp]),
    "plsr_ops_from_behaviour": (c_i32, [c_vp, c_i32, c_vp, c_vp, ctypes.POINTER(Layout), c_vp, c_vp]),
    "plsr_batch_workspace_bytes": (c_sz, [ctypes.POINTER(Layout), c_i64, c_i32]),
    "plsr_batch_plan": (c_i32, [ctypes.POINTER(Layout), c_i64, c_i32, c_i32, ctypes.POINTER(c_i32 * 4)]),
    "plsr_perm_batch": (c_i32, [c_vp, c_i64, c_i64, c_vp, ctypes.POINTER(Layout), c_vp, c_vp,
                                c_sz, c_vp]),
    "plsr_boot_batch": (c_i32, [c_vp, c_i64, c_i64, c_vp, ctypes.POINTER(Layout), c_vp, c_vp,
                                c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "plsr_boot_finalize": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp]),
    "plsr_scale_cols": (c_i32, [c_vp, c_i64, c_i32, c_vp, c_vp, c_vp]),
    "plsr_rows_frag_elems": (c_i64, [c_i32, c_i32, c_i32]),
    "plsr_ops_pack_rows": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "plsr_gram_workspace_bytes": (c_sz, [c_i32, c_i32, c_i32, c_i64, c_i64]),
    "plsr_gram_batch": (c_i32, [c_vp, c_i64, c_i64, c_i64, c_i32, c_vp, c_i32, c_i32, c_vp, c_vp, c_sz, c_vp]),
    "plsr_gather_zscore": (c_i32, [c_vp, c_i64, c_i64, c_vp, c_i32, c_i32, c_vp, c_vp, c_i32, c_vp, c_i64, c_vp]),
    "plsr_eigh_batch": (c_i32, [c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp]),
    "plsr_svd_finish": (c_i32, [c_vp, c_vp, c_i32, c_i32, ctypes.c_double, ctypes.c_double, c_vp, c_vp, c_vp]),
    "plsr_rotate_rows": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "plsr_item_fused_workspace_bytes": (c_sz, [c_i32, c_i32, c_i32, c_vp, c_i32, c_i32, c_i64, c_i32, c_i32]),
    "plsr_item_fused": (c_i32, [c_vp, c_i64, c_i64, c_i32, c_vp, c_i32, c_vp, c_vp, c_i32, c_vp, c_i32, c_i32,
                                c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i32, c_vp, c_sz, c_vp]),
    "plsr_item_agg_workspace_bytes": (c_sz, [c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i64, c_i32,
                                             c_i32]),
    "plsr_item_agg": (c_i32, [c_vp, c_i64, c_i64, c_i32, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_i32,
                              c_i32, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_sz, c_vp]),
    "plsr_item_beh_workspace_bytes": (c_sz, [c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_i32, c_i32, c_i64, c_i32]),
    "plsr_item_beh": (c_i32, [c_vp, c_i64, c_i64, c_i32, c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp, c_i32, c_vp,
                              c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_vp, c_sz, c_vp]),
    "plsr_gram_fused_workspace_bytes": (c_sz, [c_i32, c_i32, c_i32, c_vp, c_i32, c_i32, c_i64]),
    "plsr_gram_fused": (c_i32, [c_vp, c_i64, c_i64, c_i32, c_vp, c_i32, c_vp, c_vp, c_i32, c_vp, c_i32, c_i32,
                                c_vp, c_vp, c_sz, c_vp]),
    "plsr_split_gram_workspace_bytes": (c_sz, [c_i32, c_i64, c_i64, c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32]),
    "plsr_split_gram": (c_i32, [c_vp, c_i64, c_i64, c_i32, c_vp, c_vp, c_i32, c_vp, c_i32, c_vp, c_i32, c_i32, c_vp,
                                c_i32, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "plsr_split_rows_workspace_bytes": (c_sz, [c_i32, c_i64, c_i64, c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32]),
    "plsr_split_rows": (c_i32, [c_vp, c_i64, c_i64, c_i32, c_vp, c_vp, c_i32, c_vp, c_i32, c_vp, c_i32, c_i32, c_vp,
                                c_i32, c_vp, c_vp, c_i32, c_i32, c_vp, c_i64, c_vp, c_i64, c_vp, c_sz, c_vp]),
    "plsr_rows_project_workspace_bytes": (c_sz, [c_i32, c_i32, c_i32, c_i64, c_i32]),
    "plsr_rows_project": (c_i32, [c_vp, c_i64, c_i64, c_i32, c_i32, c_vp, c_i64, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp,
                                  c_vp, c_sz, c_vp]),
    "plsr_apply_rows": (c_i32, [c_vp, c_i64, c_i64, c_i32, c_vp, c_i32, c_vp, c_i64, c_vp]),
    "plsr_scale_project_rows": (c_i32, [c_vp, c_vp, c_i64, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "plsr_latent_workspace_bytes": (c_sz, [c_i32, c_i32, c_i32, c_i64]),
    "plsr_latent": (c_i32, [c_vp, c_i64, c_i64, c_i32, c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp, c_sz,
                            c_vp]),
    "plsr_mask_indices_workspace_bytes": (c_sz, [c_i64]),
    "plsr_mask_indices": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "plsr_mask_apply_rows": (c_i32, [c_vp, c_i32, c_i64, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp]),
    "plsr_latent_xt_bytes": (c_sz, [c_i32, c_i64]),
    "plsr_latent_xt_prepare": (c_i32, [c_vp, c_i64, c_i64, c_i32, c_vp, c_vp]),
    "plsr_latent_xt_workspace_bytes": (c_sz, [c_i32, c_i32, c_i32, c_i64]),
    "plsr_latent_xt": (c_i32, [c_vp, c_i64, c_i32, c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "plsr_latent_index_workspace_bytes": (c_sz, [c_i32, c_i32, c_i32, c_i64, c_i32, c_i32, c_i32]),
    "plsr_latent_xb_bytes": (c_sz, [c_i32, c_i64]),
    "plsr_latent_xb_prepare": (c_i32, [c_vp, c_i64, c_i64, c_i32, c_vp, c_vp]),
    "plsr_latent_index": (c_i32, [c_vp, c_i64, c_i32, c_vp, c_i64, c_i32, c_i32, c_i32, c_vp, c_i32, c_i32, c_vp, c_i64,
                                  c_i32, c_vp, c_i32, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "plsr_rng_permutation": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_vp]),
    "plsr_rng_permutation_seq": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_vp]),
    "plsr_rng_task_permutations": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "plsr_rng_bootstraps": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "plsr_rng_mb_permutations": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "plsr_rng_mb_bootstraps": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_i32, c_i32, c_vp, c_vp]),
    "plsr_set_tail_stream": (c_i32, [c_vp]),
    "plsr_timing_enable": (c_i32, [c_i32]),
    "plsr_timing_collect": (c_i32, [c_vp, c_vp, c_i32]),
}

_lib = None


class PlsrError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle with typed entry points."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("PLSR_LIB", _build.LIB)   # PLSR_LIB: developer override (ablation builds)
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  plspy_amd has no CPU fallback.")
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        lib = load()
        msg = lib.plsr_strerror(rc).decode()
        raise PlsrError(f"{what}: {msg} (code {rc}, hip error {lib.plsr_last_hip_error()})")
