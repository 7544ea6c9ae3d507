"""Device-side driver of the resampling kernels.

``ProjectionEngine`` keeps X resident in HBM as a torch fp64 tensor and pushes
batches of resamples through the C-ABI library (include/plsr.h) on torch's
current HIP stream.  torch is used for memory, streams and (in dist.py)
torch.distributed only; every arithmetic step on the resampling path is a
hand-written gfx950 kernel behind the ABI."""
import contextlib
import ctypes
import os

import numpy as np
import torch

from . import _lib


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# scratch one batch may use (bytes); PLSR_WORK_LIMIT_GIB overrides the default
DEFAULT_WORK_LIMIT = int(float(os.environ.get("PLSR_WORK_LIMIT_GIB", "24")) * (1 << 30))

# host arrays of this size range are uploaded through recycled page-locked staging buffers (ProjectionEngine.dev);
# PLSR_PIN_UPLOADS=0 switches that off (measurement knob of microbench/cfg3_perm_bimodal2.py)
PIN_UPLOAD_MIN, PIN_UPLOAD_MAX = ((64 << 10), (8 << 20)) if os.environ.get("PLSR_PIN_UPLOADS", "1") != "0" else (1, 0)

_SIDE_STREAMS = {}
_STAGING = {}          # (device, size class) -> list of [page-locked byte buffer, event of its last upload]


def _staged_upload(a, dtype, device):
    """Device copy (as `dtype`) of the NumPy array `a` through a page-locked staging buffer, enqueued on the
    current stream.  The buffers are a process-wide ring per power-of-two size class, reused once the upload
    that last read them has finished (an event), so that after the first few calls nothing is page-locked
    any more -- allocating a page-locked buffer is itself a page-table update (tens of milliseconds for MBs).
    The host copy into the buffer is NumPy's (one thread: torch's copy_ spins up its OpenMP team, which
    under the per-batch host work of the rb bootstrap cost 10 ms per call)."""
    np_dtype = torch.empty(0, dtype=dtype).numpy().dtype
    nbytes = a.size * np_dtype.itemsize
    cls = 1 << max(nbytes - 1, 1).bit_length()
    ring = _STAGING.setdefault((str(device), cls), [])
    slot = next((s for s in ring if s[1].query()), None)
    if slot is None:
        if len(ring) >= 8:
            slot = ring[0]
            slot[1].synchronize()
        else:
            buf = torch.empty(cls, dtype=torch.uint8, pin_memory=True)
            slot = [buf, torch.cuda.Event(), buf.numpy()]
            ring.append(slot)
    if slot is not ring[-1]:                      # least recently used first
        ring.remove(slot)
        ring.append(slot)
    np.copyto(slot[2][:nbytes].view(np_dtype).reshape(a.shape), a, casting="unsafe")
    t = slot[0][:nbytes].view(dtype).view(a.shape).to(device, non_blocking=True)
    slot[1].record(torch.cuda.current_stream())
    return t


def _side_stream(device, name):
    """The process-wide copy / tail stream `name` of a device.  Shared by every engine:
    torch's caching allocator keeps one pool per stream, so engines that each made
    their own streams (one PLS() call makes several engines) could never reuse the
    blocks of the previous call and paid a fresh device allocation (hipMalloc and the
    first touch of the new segment: 5-25 ms on this system) per call."""
    key = (str(device), name)
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return st


class _Lane:
    """Scratch of one phase kind (the permutation and bootstrap phases keep
    separate scratch so that one's reductions may overlap the other's kernel)."""

    def __init__(self):
        self.work = None
        self.frag = None


# fetch_async results of at least this many bytes do not stay in page-locked memory: a result object that keeps them
# (V, R, std_errs: 77 MB each at config 3) would otherwise keep that much page-locked host memory for as long as
# it lives.  They are downloaded straight into a pageable array whose pages a helper thread touches while the
# device still works (a copy out of the page-locked buffer cost 15 ms of config 3's 200 ms bootstrap phase, most of
# it the page faults of the fresh array).  Smaller results stay views of recycled page-locked buffers.
PINNED_COPY_BYTES = 32 << 20


class _Prefault:
    """A pageable result array whose pages are touched by a helper thread (ctypes.memset: no GIL)."""

    def __init__(self, shape, dtype):
        import threading
        self.array = np.empty(shape, dtype=dtype)
        self._th = threading.Thread(target=ctypes.memset, args=(self.array.ctypes.data, 0, self.array.nbytes))
        self._th.start()

    def ready(self):
        self._th.join()
        return self.array


class _Fetch:
    """Pending device -> host copies (ProjectionEngine.fetch_async).  Entries are either page-locked host
    tensors with their copy in flight, or (device tensor, pre-faulted pageable array) pairs that are copied
    at get() time, when the producing kernels have finished."""

    def __init__(self, host, done, stream):
        self._host, self._done, self._stream = host, done, stream

    def get(self, copy=None):
        """NumPy arrays of the fetched tensors.  Results of PINNED_COPY_BYTES or more are fresh pageable arrays;
        smaller ones are views of page-locked buffers (recycled by torch's caching host allocator once dropped)
        unless copy=True."""
        self._done.synchronize()
        out = []
        for h in self._host:
            if isinstance(h, tuple):
                t, pre = h
                dst = pre.ready()
                with torch.cuda.stream(self._stream):
                    torch.from_numpy(dst).copy_(t)         # (pageable destination: the call returns when it is done)
                out.append(dst)
                continue
            a = h.numpy()
            out.append(np.array(a) if copy else a)
        return out


class ProjectionEngine:
    """X (n x p, fp64, voxel = unit stride) on one GPU plus scratch.

    work_limit bounds the scratch a single batch may use; larger phases are
    cut into batches of resamples (each batch is one kernel launch)."""

    def __init__(self, X, device=None, work_limit=None):
        if not torch.cuda.is_available():
            raise RuntimeError("plspy_amd needs a ROCm GPU (MI355X); no CPU fallback exists")
        self.lib = _lib.load()
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        X = torch.as_tensor(X)
        if X.dim() != 2:
            raise ValueError("X must be 2-dimensional")
        self.X = X.to(device=self.device, dtype=torch.float64).contiguous()
        self.n, self.p = self.X.shape
        self.work_limit = int(work_limit) if work_limit is not None else DEFAULT_WORK_LIMIT
        self._lanes = {}
        self._pool = {}
        self._XT = None
        self._XT_n = 0
        self._XB = None
        self._XB_n = 0
        self._tail = None
        self._h2d = None
        self._d2h = None
        self.last_item_kernel = None          # which VS / latent kernels the last batch took (read by the tests)
        self.last_latent_kernel = None

    # -- helpers -----------------------------------------------------------
    def _buf(self, name, nbytes):
        """Scratch that lives with the engine (bytes, grown on demand).  The batches of a phase run
        on one stream, so they can share one buffer of each kind; allocating per batch let torch's
        caching allocator split a cached 9.6 GB block for a smaller request now and then, and the
        next batch paid a fresh hipMalloc plus the first touch of its pages (one 35 ms kernel in
        every dozen)."""
        t = self._pool.get(name)
        if t is None or t.numel() < nbytes:
            self._pool[name] = t = None
            t = self._pool[name] = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        return t

    def release_scratch(self):
        """Drops the engine-owned scratch (the pooled work / VS^T blocks -- 9.6 GB at config 3 --, the padded
        transposed copy of X, the phase lanes): call when a PLS() call's resampling is over.  The next phase
        allocates again (torch's caching allocator keeps the blocks for reuse by later engines)."""
        self._pool.clear()
        self._lanes.clear()
        self._XT = None
        self._XT_n = 0
        self._XB = None
        self._XB_n = 0

    def dev(self, a, dtype=torch.float64):
        """Device copy of a host array (or the tensor itself if it is already
        there).  Host arrays are uploaded on a dedicated copy stream: a pageable
        copy on the compute stream would block the host until every kernel
        enqueued before it has finished, so the host could never prepare batch
        i + 1 while the device works on batch i."""
        if a is None:
            return None
        if isinstance(a, np.ndarray):
            if self._h2d is None:
                self._h2d = _side_stream(self.device, "h2d")
            cur = torch.cuda.current_stream()
            staged = PIN_UPLOAD_MIN <= a.size * torch.empty(0, dtype=dtype).element_size() <= PIN_UPLOAD_MAX
            if not staged:
                src = torch.from_numpy(np.ascontiguousarray(a))
                if src.dtype != dtype:
                    src = src.to(dtype)
            with torch.cuda.stream(self._h2d):
                if staged:
                    # mid-sized per-batch tables (behaviour z-scores, operator rows, index tables) go through a
                    # recycled page-locked staging buffer (_staged_upload): copied straight from fresh
                    # pageable memory, the runtime page-locks the caller's pages for the transfer,
                    # i.e. updates the GPU's page tables under whatever kernel is running -- the first
                    # projection launch of a phase then took 35-50 ms instead of 19 (config 3's "bimodal"
                    # permutation time)
                    t = _staged_upload(a, dtype, self.device)
                else:
                    t = src.to(self.device)
            cur.wait_stream(self._h2d)
            t.record_stream(cur)
            return t
        return torch.as_tensor(a).to(device=self.device, dtype=dtype).contiguous()

    def layout(self, k, R):
        lay = _lib.Layout()
        _lib.check(self.lib.plsr_layout_init(self.n, k, R, ctypes.byref(lay)),
                   f"plsr_layout_init(n={self.n}, k={k}, R={R})")
        return lay

    def plan(self, k, R, k2=0, boot=True):
        """Launch shape of a batch of R resamples (plsr_batch_plan): dict(voxel_tiles, splits,
        tiles, register_resident)."""
        lay = self.layout(k, R)
        out = (ctypes.c_int32 * 4)()
        _lib.check(self.lib.plsr_batch_plan(ctypes.byref(lay), self.p, k2, int(boot), ctypes.byref(out)),
                   "plsr_batch_plan")
        return dict(voxel_tiles=out[0], splits=out[1], tiles=out[2], register_resident=bool(out[3]))

    def batch_size(self, k, k2, R):
        """Largest batch (a multiple of 16 resamples, or all of them) whose
        scratch fits work_limit."""
        b = int(R)
        while True:
            lay = self.layout(k, b)
            need = self.lib.plsr_batch_workspace_bytes(ctypes.byref(lay), self.p, k2) + lay.frag_elems * 8
            if need <= self.work_limit or b <= 16:
                return b
            b = max(16, (b // 2 + 15) // 16 * 16)

    def _scratch(self, lay, k2, lane=None):
        L = self._lanes.get(lane)
        if L is None:
            L = self._lanes[lane] = _Lane()
        need = self.lib.plsr_batch_workspace_bytes(ctypes.byref(lay), self.p, k2)
        if L.work is None or L.work.numel() < need:
            L.work = None
            L.work = torch.empty(need, dtype=torch.uint8, device=self.device)
        if L.frag is None or L.frag.numel() < lay.frag_elems:
            L.frag = None
            L.frag = torch.empty(lay.frag_elems, dtype=torch.float64, device=self.device)
        return L.work, L.frag, need

    # -- overlapped reduction tail ------------------------------------------
    def join(self):
        """Current stream waits for the reduction tails of bootstrap phases
        started with ``overlap_tail=True``; call before reading their results."""
        if self._tail is not None:
            torch.cuda.current_stream().wait_stream(self._tail)

    def tail_stream(self):
        """Context in which torch work (the collectives of dist.exchange) is enqueued
        on the tail stream, i.e. behind the reductions of an overlapped bootstrap
        phase and beside whatever the main stream runs next."""
        if self._tail is None:
            self._tail = _side_stream(self.device, "tail")
        return torch.cuda.stream(self._tail)

    def _build_ops(self, lay, frag, inds=None, M=None, cols=None, beh=None):
        if beh is not None:
            Yz, U, rowcell = beh
            _lib.check(self.lib.plsr_ops_from_behaviour(_ptr(Yz), int(Yz.shape[2]), _ptr(U), _ptr(rowcell),
                                                        ctypes.byref(lay), _ptr(frag), _stream()),
                       "plsr_ops_from_behaviour")
        elif cols is not None:
            _lib.check(self.lib.plsr_ops_pack(_ptr(cols), ctypes.byref(lay), _ptr(frag), _stream()),
                       "plsr_ops_pack")
        else:
            _lib.check(self.lib.plsr_ops_from_indices(_ptr(inds), _ptr(M), ctypes.byref(lay),
                                                      _ptr(frag), _stream()),
                       "plsr_ops_from_indices")

    # -- permutation ---------------------------------------------------------
    def perm_prepare(self, k, inds, M):
        """The operator fragments of a permutation phase, built NOW on a side stream: when a bootstrap phase is
        enqueued between this call and perm_phase(prepared=...), the permutation's operator kernel runs beside
        the bootstrap's instead of between the two projection kernels (20 us of a rank's 0.85 ms step at
        config 2 on eight GPUs).  The side stream starts behind everything the current stream holds so far
        (the inputs; an earlier permutation kernel reading the same fragment buffer).  Returns a handle, or
        None when the phase needs more than one batch."""
        R = int(inds.shape[0])
        if R == 0 or self.batch_size(k, 0, R) < R:
            return None
        lay = self.layout(k, R)
        work, frag, need = self._scratch(lay, 0, "perm")
        d_inds, Md = self.dev(inds, torch.int32), self.dev(M)
        cur = torch.cuda.current_stream()
        side = _side_stream(self.device, "ops")
        ev0 = torch.cuda.Event()
        ev0.record(cur)
        with torch.cuda.stream(side):
            side.wait_event(ev0)
            self._build_ops(lay, frag, inds=d_inds, M=Md)
            ready = torch.cuda.Event()
            ready.record(side)
        for t in (d_inds, Md, frag):
            t.record_stream(side)
        return dict(k=k, R=R, lay=lay, work=work, frag=frag, need=need, ready=ready)

    def perm_phase(self, k, inds=None, M=None, cols=None, beh=None, prepared=None):
        """s_hat^2 (R x k) for every resample.  Either ``inds`` (R x n int32
        row selections) with ``M`` (n x k), or dense ``cols`` (R x k x n), or
        ``beh`` = (Yz (R, n, b) per-cell z-scored behaviour, U (cells*b, k),
        rowcell (n,)) for the behaviour-PLS operator Yz_cell @ U_cell; or ``prepared``, the handle
        of perm_prepare (operators already built)."""
        if prepared is not None:
            out = torch.empty((prepared["R"], k), dtype=torch.float64, device=self.device)
            torch.cuda.current_stream().wait_event(prepared["ready"])
            _lib.check(self.lib.plsr_perm_batch(_ptr(self.X), self.X.stride(0), self.p, _ptr(prepared["frag"]),
                                                ctypes.byref(prepared["lay"]), _ptr(out), _ptr(prepared["work"]),
                                                prepared["need"], _stream()), "plsr_perm_batch")
            return out
        if beh is not None and callable(beh[0]):
            R = int(beh[3])                      # (Yz_fn(lo, hi), U, rowcell, R): Yz made batch by batch
        else:
            R = int(inds.shape[0] if inds is not None else (cols.shape[0] if cols is not None else beh[0].shape[0]))
        out = torch.empty((R, k), dtype=torch.float64, device=self.device)
        if R == 0:
            return out
        step = self.batch_size(k, 0, R)
        if beh is not None and callable(beh[0]):
            step = min(step, 512)               # several batches, so that the host's Yz of batch i+1 hides behind batch i
        Md = self.dev(M)
        for lo in range(0, R, step):
            hi = min(R, lo + step)
            lay = self.layout(k, hi - lo)
            work, frag, need = self._scratch(lay, 0, "perm")
            if beh is not None:
                Yz = beh[0](lo, hi) if callable(beh[0]) else beh[0][lo:hi]
                self._build_ops(lay, frag, beh=(self.dev(Yz), self.dev(beh[1]), self.dev(beh[2], torch.int32)))
            elif cols is not None:
                self._build_ops(lay, frag, cols=self.dev(cols[lo:hi]))
            else:
                self._build_ops(lay, frag, inds=self.dev(inds[lo:hi], torch.int32), M=Md)
            _lib.check(self.lib.plsr_perm_batch(_ptr(self.X), self.X.stride(0), self.p, _ptr(frag),
                                                ctypes.byref(lay), _ptr(out[lo:hi]), _ptr(work),
                                                need, _stream()), "plsr_perm_batch")
        return out

    # -- bootstrap -----------------------------------------------------------
    def boot_phase(self, k, inds=None, M=None, cols=None, ref=None, Xm=None, dump=False,
                   overlap_tail=False):
        """Streams a bootstrap phase.  Returns dict(S1, S2 (p x k shifted
        moments), ssq (R x k), T (R x k x k2) or None, vs (R x p x k) or None).
        With ``overlap_tail`` the slab reductions that end each batch run on a
        side stream (so the next phase's kernel overlaps them); ``join()`` before
        reading the results."""
        R = int(inds.shape[0] if inds is not None else cols.shape[0])
        refd = self.dev(ref)
        Xmd = self.dev(Xm)
        k2 = 0 if Xmd is None else int(Xmd.shape[0])
        if overlap_tail and self._tail is None:
            self._tail = _side_stream(self.device, "tail")
        if overlap_tail and R:
            # the moment block is first touched by the merges on the tail stream: zeroed there, the fill (19 MB,
            # 11 us at config 2) runs beside the operator kernels instead of in front of the projection
            self._tail.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._tail):
                S12 = torch.zeros((2, self.p, k), dtype=torch.float64, device=self.device)
            S12.record_stream(torch.cuda.current_stream())
        else:
            S12 = torch.zeros((2, self.p, k), dtype=torch.float64, device=self.device)   # one block: one all_reduce
        S1, S2 = S12[0], S12[1]
        ssq = torch.empty((R, k), dtype=torch.float64, device=self.device)
        T = torch.empty((R, k, k2), dtype=torch.float64, device=self.device) if k2 else None
        vs = torch.zeros((R, self.p, k), dtype=torch.float64, device=self.device) if dump else None
        if R:
            step = self.batch_size(k, k2, R)
            Md = self.dev(M)
            if not overlap_tail:
                self.join()          # an earlier overlapped tail may still read the bootstrap scratch
            tail = ctypes.c_void_p(self._tail.cuda_stream) if overlap_tail else ctypes.c_void_p(0)
            _lib.check(self.lib.plsr_set_tail_stream(tail), "plsr_set_tail_stream")
            try:
                for lo in range(0, R, step):
                    hi = min(R, lo + step)
                    lay = self.layout(k, hi - lo)
                    work, frag, need = self._scratch(lay, k2, "boot")
                    if cols is not None:
                        self._build_ops(lay, frag, cols=self.dev(cols[lo:hi]))
                    else:
                        self._build_ops(lay, frag, inds=self.dev(inds[lo:hi], torch.int32), M=Md)
                    _lib.check(self.lib.plsr_boot_batch(
                        _ptr(self.X), self.X.stride(0), self.p, _ptr(frag), ctypes.byref(lay),
                        _ptr(refd), _ptr(Xmd), Xmd.stride(0) if k2 else 0, k2,
                        _ptr(S1), _ptr(S2), _ptr(ssq[lo:hi]), _ptr(T[lo:hi]) if k2 else _ptr(None),
                        _ptr(vs[lo:hi]) if dump else _ptr(None), _ptr(work), need, _stream()),
                        "plsr_boot_batch")
            finally:
                self.lib.plsr_set_tail_stream(ctypes.c_void_p(0))
        return {"S1": S1, "S2": S2, "S12": S12, "ssq": ssq, "T": T, "vs": vs, "R": R}

    def boot_finalize(self, S1, S2, R, num=None):
        """(std_errs, boot_ratios) from summed shifted moments."""
        sd = torch.empty_like(S1)
        ratio = torch.empty_like(S1) if num is not None else None
        numd = self.dev(num)
        _lib.check(self.lib.plsr_boot_finalize(_ptr(S1), _ptr(S2), _ptr(numd), S1.numel(), int(R),
                                               _ptr(sd), _ptr(ratio), _stream()),
                   "plsr_boot_finalize")
        return sd, ratio

    def scale_cols(self, M, scale):
        """M (rows x cols, host or device) times scale per column, on the device: the observed V s
        from V and s.  A host array's device copy is scaled in place; a device tensor is left alone."""
        if isinstance(M, np.ndarray) and M.ndim == 2 and not M.flags["C_CONTIGUOUS"] and M.T.flags["C_CONTIGUOUS"]:
            # V as the PLS classes return it: the transpose of the (k, p) block the device produced.  Its bytes go up as
            # they lie and the layout copy is made on the device (a strided host copy of 77 MB took 20 ms at config 3)
            Md = self.dev(M.T).t().contiguous()
        else:
            Md = self.dev(M).contiguous()
        out = Md if isinstance(M, np.ndarray) else torch.empty_like(Md)
        sd = self.dev(np.ascontiguousarray(scale, dtype=np.float64) if isinstance(scale, np.ndarray) else scale)
        if Md.dim() != 2 or sd.numel() != Md.shape[1]:
            raise ValueError("scale_cols: one scale per column")
        _lib.check(self.lib.plsr_scale_cols(_ptr(Md), Md.shape[0], Md.shape[1], _ptr(sd), _ptr(out), _stream()),
                   "plsr_scale_cols")
        return out

    def apply_operator(self, rows):
        """(m x n) operator rows -> (m x p) = rows @ X on the device (K0; the observed
        cell means / centred block / correlation block / back-projection)."""
        d_rows = rows.contiguous() if torch.is_tensor(rows) else self.dev(np.ascontiguousarray(rows, dtype=np.float64))
        m, n = d_rows.shape
        if n != self.n:
            raise ValueError(f"operator rows have {n} columns, X has {self.n} rows")
        out = torch.empty((m, self.p), dtype=torch.float64, device=self.device)
        _lib.check(self.lib.plsr_apply_rows(_ptr(self.X), self.X.stride(0), self.p, self.n, _ptr(d_rows), m,
                                            _ptr(out), self.p, _stream()), "plsr_apply_rows")
        return out

    def latent_batch(self, vst, n, Zt, nsq=None):
        """K5 / K5x for a batch: Zt (cnt, k, n) <- VS_b X[:n]^T from vst (cnt, k, p), nsq (cnt, k) the
        squared column norms (optional).  With at most 128 data rows X is read as pre-transposed B
        fragments (made once per engine); otherwise through the LDS-staged kernel."""
        cnt, k = int(vst.shape[0]), int(vst.shape[1])
        if n <= 128 and k <= 64 and os.environ.get("PLSR_LATENT_XT", "1") != "0":
            need = self.lib.plsr_latent_xt_workspace_bytes(n, k, cnt, self.p)
            work = self._buf("k5work", need)
            _lib.check(self.lib.plsr_latent_xt(_ptr(self._xt(n)), self.p, n, _ptr(vst), vst.stride(1), cnt, k, _ptr(Zt),
                                               _ptr(nsq), _ptr(work), need, _stream()), "plsr_latent_xt")
            return
        need = self.lib.plsr_latent_workspace_bytes(n, k, cnt, self.p)
        if need == 0:
            raise _lib.PlsrError(f"plsr_latent: unsupported shape n={n} k={k}")
        work = self._buf("k5work", need)
        _lib.check(self.lib.plsr_latent(_ptr(self.X), self.X.stride(0), self.p, n, _ptr(vst), vst.stride(1), cnt, k,
                                        _ptr(Zt), _ptr(nsq), _ptr(work), need, _stream()), "plsr_latent")

    def _xt(self, n):
        if self._XT is None or self._XT_n != n:
            nb = self.lib.plsr_latent_xt_bytes(n, self.p)
            self._XT = torch.empty(nb // 8, dtype=torch.float64, device=self.device)
            _lib.check(self.lib.plsr_latent_xt_prepare(_ptr(self.X), self.X.stride(0), self.p, n, _ptr(self._XT),
                                                       _stream()), "plsr_latent_xt_prepare")
            self._XT_n = n
        return self._XT

    def _xb(self, n):
        """Tile-major copy of X[:n] (K5i's operand), made once per engine."""
        if self._XB is None or self._XB_n != n:
            nb = self.lib.plsr_latent_xb_bytes(n, self.p)
            self._XB = torch.empty(nb // 8, dtype=torch.float64, device=self.device)
            _lib.check(self.lib.plsr_latent_xb_prepare(_ptr(self.X), self.X.stride(0), self.p, n, _ptr(self._XB),
                                                       _stream()), "plsr_latent_xb_prepare")
            self._XB_n = n
        return self._XB

    @staticmethod
    def distinct_rows(idx):
        """Largest number of different entries in a row of idx (cnt, m)."""
        srt = np.sort(idx, axis=1)
        return int((np.count_nonzero(srt[:, 1:] != srt[:, :-1], axis=1) + 1).max())

    def index_served(self, n, k, cnt, m, most, t_rows=0):
        """Whether K5i serves a batch of cnt samples of m rows with at most `most` different ones."""
        return bool(most) and n <= 128 and \
            self.lib.plsr_latent_index_workspace_bytes(n, k, cnt, self.p, m, most, t_rows) > 0

    def latent_batch_index(self, vst, n, idx, d_idx, L, nsq=None, tiled=False, most=None, own=None):
        """K5i: L (cnt, k, m) <- (X[idx_b] VS_b^T)^T, the latent scores of every sample's own rows
        (`_compute_X_latents(X_new, V_hat)` before the normalisation), computed on the rows of X the sample
        holds, each once (plsr_latent_index).  idx (cnt, m): host copy of d_idx (int32, rows of X[:n]).
        tiled: vst is tile-major (item_beh(tiled=True); only K5i reads that layout).
        own = (T (cnt, rows, ld) device block, row indices): t further columns L[b, j, m + t] = VS_b[j] . T_b[row_t]
        (products with rows of the item's own block -- the multiblock's raw task rows; L is (cnt, k, m + t)).
        Shapes the library does not serve (n > 128): the full product and a gather of its columns; `own` is then
        refused (the caller asks index_served first)."""
        cnt, k = int(vst.shape[0]), int(vst.shape[1])
        m = int(idx.shape[1])
        t_rows = 0 if own is None else len(own[1])
        if most is None:
            most = self.distinct_rows(idx) if n <= 128 else 0
        if self.index_served(n, k, cnt, m, most, t_rows):
            need = self.lib.plsr_latent_index_workspace_bytes(n, k, cnt, self.p, m, most, t_rows)
            work = self._buf("k5work", need)
            T = d_tr = None
            if own is not None:
                T = own[0]
                d_tr = self.dev(np.asarray(own[1], dtype=np.int32), torch.int32)
            _lib.check(self.lib.plsr_latent_index(_ptr(self._xb(n)), self.p, n, _ptr(vst), vst.stride(1), int(tiled),
                                                  cnt, k, _ptr(d_idx), m, most, _ptr(T),
                                                  0 if T is None else T.stride(1), 0 if T is None else int(T.shape[1]),
                                                  _ptr(d_tr), t_rows, _ptr(L), _ptr(nsq), _ptr(work), need, _stream()),
                       "plsr_latent_index")
            self.last_latent_kernel = "index"
            return
        if tiled or own is not None:
            raise _lib.PlsrError("tile-major VS^T / own-row products are only served by plsr_latent_index")
        Zt = torch.empty((cnt, k, n), dtype=torch.float64, device=self.device)
        self.latent_batch(vst, n, Zt, nsq)
        torch.gather(Zt, 2, d_idx.long()[:, None, :].expand(cnt, k, m), out=L)
        self.last_latent_kernel = "full"

    def latents_device(self, vt):
        """(1, k, n) device tensor (X @ V)^T for V^T = vt (k, p) on the device (K5, one item)."""
        k = int(vt.shape[0])
        need = self.lib.plsr_latent_workspace_bytes(self.n, k, 1, self.p)
        if need == 0:
            raise _lib.PlsrError(f"plsr_latent: unsupported shape n={self.n} k={k}")
        vt = vt.contiguous()
        Zt = torch.empty((1, k, self.n), dtype=torch.float64, device=self.device)
        work = torch.empty(need, dtype=torch.uint8, device=self.device)
        _lib.check(self.lib.plsr_latent(_ptr(self.X), self.X.stride(0), self.p, self.n, _ptr(vt), self.p, 1, k,
                                        _ptr(Zt), _ptr(None), _ptr(work), need, _stream()), "plsr_latent")
        return Zt

    def latents(self, V, rows=None):
        """X @ V (n x k) for a host matrix V (p x k): the observed latent scores
        (class_functions.py:165-182, pls_classes.py:263) and the bootstrap's
        left_sv_sampled base `X @ V` (bootstrap_permutation.py:617), through K5 with a single
        item.  On the host this product reads all of X (5 ms at config 2, 40 ms at config 5)."""
        V = np.asarray(V, dtype=np.float64)
        if V.ndim != 2 or V.shape[0] != self.p:
            raise ValueError(f"V must be ({self.p}, k)")
        Zt = self.latents_device(self.dev(V).t())           # (k, p): a layout copy on the device
        return np.ascontiguousarray(Zt[0].t().cpu().numpy())

    # -- results to the host without stalling the pipeline --------------------------
    def fetch_async(self, tensors):
        """Start copying device tensors into page-locked host memory on the download stream
        (behind everything enqueued on the current stream so far).  Returns a handle whose
        ``get()`` waits for these copies only and returns NumPy arrays (views of the pinned
        buffers, which torch's caching host allocator recycles once the arrays are dropped; tensors of
        PINNED_COPY_BYTES or more: fresh pageable arrays, pre-faulted meanwhile, see _Fetch).
        A `.cpu()` instead would block the host until the whole current stream has drained and
        move the data through a pageable staging copy."""
        if self._d2h is None:
            self._d2h = _side_stream(self.device, "d2h")
        cur = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(cur)
        host = []
        with torch.cuda.stream(self._d2h):
            self._d2h.wait_event(ev)
            for t in tensors:
                t.record_stream(self._d2h)
                if t.numel() * t.element_size() >= PINNED_COPY_BYTES:
                    host.append((t, _Prefault(tuple(t.shape), torch.empty(0, dtype=t.dtype).numpy().dtype)))
                    continue
                h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                h.copy_(t, non_blocking=True)
                host.append(h)
            done = torch.cuda.Event()
            done.record(self._d2h)
        return _Fetch(host, done, self._d2h)

    # -- K2: Gram / thin SVD -------------------------------------------------
    def gram_phase(self, rows, gather=None):
        """rows: (S, m, n') operator rows (NumPy).  Returns the (S, mm, mm) Grams
        (A_s Z_s)(A_s Z_s)^T on the device, mm = 16*ceil(m/16).

        gather=None: Z_s = X for every item (n' = n).  Otherwise gather is a
        dict(src=(S, n') int32, cell_lo, cell_z) and Z_s is the item's own
        gathered / per-cell z-scored copy of X (gather_zscore), made in chunks."""
        if not torch.is_tensor(rows):                    # (a device tensor: rows formed on the device)
            rows = np.ascontiguousarray(rows, dtype=np.float64)
        S, m, n = rows.shape
        assert n == (self.n if gather is None else gather["src"].shape[1])
        mm = (m + 15) // 16 * 16
        G = torch.empty((S, mm, mm), dtype=torch.float64, device=self.device)
        per_item = self.lib.plsr_rows_frag_elems(n, m, 1) * 8 + m * n * 8
        if gather is not None:
            ncell = len(gather["cell_z"])
            lo_c = (ctypes.c_int32 * (ncell + 1))(*[int(x) for x in gather["cell_lo"]])
            z_c = (ctypes.c_int32 * ncell)(*[int(x) for x in gather["cell_z"]])
            per_item += 2 * ncell * self.p * 8                 # per-(item, cell, voxel) scale and shift
        step = max(1, min(S, self.work_limit // max(per_item, 1)))
        for lo in range(0, S, step):
            hi = min(S, lo + step)
            cnt = hi - lo
            d_rows = self.dev(rows[lo:hi])
            frag = torch.empty(self.lib.plsr_rows_frag_elems(n, m, cnt), dtype=torch.float64,
                               device=self.device)
            _lib.check(self.lib.plsr_ops_pack_rows(_ptr(d_rows), cnt, m, n, _ptr(frag), _stream()),
                       "plsr_ops_pack_rows")
            if gather is None:
                need = self.lib.plsr_gram_workspace_bytes(n, m, cnt, self.p, 0)
                if need == 0:
                    raise _lib.PlsrError(f"plsr_gram: unsupported shape n={n} m={m}")
                work = torch.empty(need, dtype=torch.uint8, device=self.device)
                _lib.check(self.lib.plsr_gram_batch(_ptr(self.X), 0, self.X.stride(0), self.p, n, _ptr(frag),
                                                    cnt, m, _ptr(G[lo:hi]), _ptr(work), need, _stream()),
                           "plsr_gram_batch")
            else:
                # the item matrices (gathered, per-cell z-scored rows of X) exist only
                # inside the kernel: statistics pass + Gram with the z-score fused in
                d_src = self.dev(np.ascontiguousarray(gather["src"][lo:hi]), torch.int32)
                need = self.lib.plsr_gram_fused_workspace_bytes(self.n, n, m, lo_c, ncell, cnt, self.p)
                if need == 0:
                    raise _lib.PlsrError(f"plsr_gram_fused: unsupported shape n={n} m={m} ncell={ncell}")
                work = torch.empty(need, dtype=torch.uint8, device=self.device)
                _lib.check(self.lib.plsr_gram_fused(_ptr(self.X), self.X.stride(0), self.p, self.n, _ptr(d_src), n,
                                                    lo_c, z_c, ncell, _ptr(frag), cnt, m, _ptr(G[lo:hi]),
                                                    _ptr(work), need, _stream()), "plsr_gram_fused")
        return G

    def split_gram(self, cells, Y):
        """K2s (plsr_split_gram): the per-item Grams of behaviour / multiblock split-half items from their
        cell description (split_half_resampling._cells_rb / _cells_mb), or None when the kernel's instances do
        not serve the shape (the caller then takes gram_phase's fused path).  Returns (G (S, mm, mm), rownorm
        (S, mm) or None) on the device; when cells["normalise"] the multiblock row normalisation is applied to
        G and rownorm holds the norms of the un-normalised rows."""
        xsrc = np.ascontiguousarray(cells["xsrc"], dtype=np.int32)
        ysrc = np.ascontiguousarray(cells["ysrc"], dtype=np.int32)
        S, nz = xsrc.shape
        rows = [int(x) for x in cells["cell_rows"]]
        nq, nbq = len(rows), int(cells["nbq"])
        Wc = cells.get("Wc")
        ktask = 0 if Wc is None else int(Wc.shape[0])
        m = len(cells["row_cell"])
        Yd = self.dev(np.ascontiguousarray(Y, dtype=np.float64))
        b = int(Yd.shape[1])
        c_rows = (ctypes.c_int32 * nq)(*rows)
        r_cell = (ctypes.c_int32 * m)(*[int(x) for x in cells["row_cell"]])
        r_sub = (ctypes.c_int32 * m)(*[int(x) for x in cells["row_sub"]])
        ldx = self.X.stride(0)
        if not self.lib.plsr_split_gram_workspace_bytes(self.n, ldx, self.p, b, c_rows, nq, nbq, ktask, m, 1):
            return None
        mm = (m + 15) // 16 * 16
        G = torch.empty((S, mm, mm), dtype=torch.float64, device=self.device)
        norm = bool(cells.get("normalise"))
        rown = torch.empty((S, mm), dtype=torch.float64, device=self.device) if norm else None
        Wd = self.dev(np.ascontiguousarray(Wc, dtype=np.float64)) if ktask else None
        per_item = self.lib.plsr_split_gram_workspace_bytes(self.n, ldx, self.p, b, c_rows, nq, nbq, ktask, m, 64) // 64
        step = max(1, min(S, self.work_limit // max(per_item, 1)))
        for lo in range(0, S, step):
            hi = min(S, lo + step)
            cnt = hi - lo
            need = self.lib.plsr_split_gram_workspace_bytes(self.n, ldx, self.p, b, c_rows, nq, nbq, ktask, m, cnt)
            work = self._buf("k2work", need)
            d_x = self.dev(xsrc[lo:hi], torch.int32)
            d_y = self.dev(ysrc[lo:hi], torch.int32)
            _lib.check(self.lib.plsr_split_gram(
                _ptr(self.X), ldx, self.p, self.n, _ptr(d_x), _ptr(d_y), nz, _ptr(Yd), b, c_rows, nq, nbq, _ptr(Wd),
                ktask, r_cell, r_sub, m, int(norm), cnt, _ptr(G[lo:hi]), _ptr(rown[lo:hi]) if norm else _ptr(None),
                _ptr(work), need, _stream()), "plsr_split_gram")
        return G, rown

    def split_rows(self, cells, Y, pool=False):
        """The ROWS variant of K2s (plsr_split_rows): the un-normalised cross-block rows of every item, (items, m, p)
        in logical row order, and their squared norms (items, m16) -- from the same cell description as
        split_gram.  None when the kernel's instances do not serve the shape."""
        xsrc = np.ascontiguousarray(cells["xsrc"], dtype=np.int32)
        ysrc = np.ascontiguousarray(cells["ysrc"], dtype=np.int32)
        items, nz = xsrc.shape
        rows = [int(x) for x in cells["cell_rows"]]
        nq, nbq = len(rows), int(cells["nbq"])
        Wc = cells.get("Wc")
        ktask = 0 if Wc is None else int(Wc.shape[0])
        m = len(cells["row_cell"])
        Yd = self.dev(np.ascontiguousarray(Y, dtype=np.float64))
        b = int(Yd.shape[1])
        c_rows = (ctypes.c_int32 * nq)(*rows)
        r_cell = (ctypes.c_int32 * m)(*[int(x) for x in cells["row_cell"]])
        r_sub = (ctypes.c_int32 * m)(*[int(x) for x in cells["row_sub"]])
        ldx = self.X.stride(0)
        need = self.lib.plsr_split_rows_workspace_bytes(self.n, ldx, self.p, b, c_rows, nq, nbq, ktask, m, items)
        if not need:
            return None
        R = self._vst(items, m, pool)
        m16 = (m + 15) // 16 * 16
        rowsq = torch.zeros((items, m16), dtype=torch.float64, device=self.device)
        Wd = self.dev(np.ascontiguousarray(Wc, dtype=np.float64)) if ktask else None
        work = self._buf("k2work", need)
        d_x, d_y = self.dev(xsrc, torch.int32), self.dev(ysrc, torch.int32)
        _lib.check(self.lib.plsr_split_rows(
            _ptr(self.X), ldx, self.p, self.n, _ptr(d_x), _ptr(d_y), nz, _ptr(Yd), b, c_rows, nq, nbq, _ptr(Wd), ktask,
            r_cell, r_sub, m, items, _ptr(R), self.p, _ptr(rowsq), m16, _ptr(work), need, _stream()), "plsr_split_rows")
        return R, rowsq

    def gather_zscore(self, src, cell_lo, cell_z):
        """(items, nout, p) tensor: rows of X gathered by src (items x nout) and
        z-scored (ddof 0, / sqrt(n_cell)) within the output-row cells flagged in
        cell_z; other cells are plain copies.  The data side of _compute_corr."""
        src = np.ascontiguousarray(np.atleast_2d(src), dtype=np.int32)
        items, nout = src.shape
        cell_lo = np.ascontiguousarray(cell_lo, dtype=np.int32)
        cell_z = np.ascontiguousarray(cell_z, dtype=np.int32)
        assert cell_lo[0] == 0 and cell_lo[-1] == nout and len(cell_z) == len(cell_lo) - 1
        out = torch.empty((items, nout, self.p), dtype=torch.float64, device=self.device)
        # keep the small device arrays referenced until the launch is enqueued:
        # a temporary freed between two allocations would be reused by the next
        d_src = self.dev(src, torch.int32)
        d_lo = self.dev(cell_lo, torch.int32)
        d_z = self.dev(cell_z, torch.int32)
        _lib.check(self.lib.plsr_gather_zscore(
            _ptr(self.X), self.X.stride(0), self.p, _ptr(d_src), items, nout, _ptr(d_lo), _ptr(d_z),
            len(cell_z), _ptr(out), self.p, _stream()), "plsr_gather_zscore")
        return out

    # -- K4 / K5: bootstrap with per-resample matrices --------------------------
    @staticmethod
    def source_ranges(src, cell_lo):
        """(src_lo, src_hi): per cell, the range of source rows the items of `src` (items, nz;
        NumPy) read -- what plsr_item_agg needs to know (a bootstrap draws a cell's rows from
        the cell's own subjects, resample.py:132-160)."""
        src = np.asarray(src)
        lo = np.array([src[:, a:b].min() for a, b in zip(cell_lo[:-1], cell_lo[1:])], dtype=np.int32)
        hi = np.array([src[:, a:b].max() + 1 for a, b in zip(cell_lo[:-1], cell_lo[1:])], dtype=np.int32)
        return lo, hi

    def _agg_bytes(self, nz, k, cell_lo, cell_z, ranges, items, moments, rowsq):
        """Workspace of plsr_item_agg for this shape, 0 when it does not serve it."""
        ncell = len(cell_z)
        lo = (ctypes.c_int32 * (ncell + 1))(*[int(x) for x in cell_lo])
        zf = (ctypes.c_int32 * ncell)(*[int(x) for x in cell_z])
        slo = (ctypes.c_int32 * ncell)(*[int(x) for x in ranges[0]])
        shi = (ctypes.c_int32 * ncell)(*[int(x) for x in ranges[1]])
        return self.lib.plsr_item_agg_workspace_bytes(self.n, nz, k, lo, zf, slo, shi, ncell, items, self.p,
                                                      int(moments), int(rowsq))

    def item_fused(self, src, cell_lo, cell_z, rows, ref=None, S1=None, S2=None, want_vst=False,
                   want_rowsq=False, stats=None, src_ranges=None, pool=False):
        """K4a / K4f: VS_b = rows_b @ Z_b for every item without materialising Z_b
        (Z_b = X[src_b] z-scored within the cells flagged in cell_z).
        rows (items, k, nz).  S1 / S2 (p, k) are accumulated into when given.
        src_ranges: None, or (src_lo, src_hi) -- the source-row range of every cell
        (source_ranges); the aggregated-operator kernel (plsr_item_agg) then serves when the
        shape allows, the gathering kernel (plsr_item_fused) otherwise.
        stats: None, or a dict shared by several calls of the gathering kernel on the SAME src /
        cells: the first call stores the per-(item, cell, voxel) scale and shift in it, later
        calls reuse them.
        Returns (vst (items, k, p) or None, rowsq (items, k) or None)."""
        d_src = self.dev(src, torch.int32)
        d_rows = self.dev(rows)
        items, k, nz = d_rows.shape
        assert d_src.shape == (items, nz)
        lo = (ctypes.c_int32 * (len(cell_lo)))(*[int(x) for x in cell_lo])
        zf = (ctypes.c_int32 * (len(cell_z)))(*[int(x) for x in cell_z])
        ncell = len(cell_z)
        if src_ranges is not None:
            slo = (ctypes.c_int32 * ncell)(*[int(x) for x in src_ranges[0]])
            shi = (ctypes.c_int32 * ncell)(*[int(x) for x in src_ranges[1]])
            need = self.lib.plsr_item_agg_workspace_bytes(self.n, nz, k, lo, zf, slo, shi, ncell, items, self.p,
                                                          int(S1 is not None), int(want_rowsq))
            if need:
                self.last_item_kernel = "agg"
                work = self._buf("k4work", need) if pool else torch.empty(need, dtype=torch.uint8, device=self.device)
                refd = self.dev(ref)
                vst = self._vst(items, k, pool) if want_vst else None
                k16 = (k + 15) // 16 * 16
                rowsq = torch.empty((items, k16), dtype=torch.float64, device=self.device) if want_rowsq else None
                _lib.check(self.lib.plsr_item_agg(
                    _ptr(self.X), self.X.stride(0), self.p, self.n, _ptr(d_src), nz, lo, zf, slo, shi, ncell,
                    _ptr(d_rows), items, k, _ptr(refd), _ptr(S1), _ptr(S2), _ptr(vst), self.p, _ptr(rowsq),
                    _ptr(work), need, _stream()), "plsr_item_agg")
                return vst, (rowsq[:, :k] if want_rowsq else None)
        self.last_item_kernel = "gather"
        need = self.lib.plsr_item_fused_workspace_bytes(self.n, nz, k, lo, ncell, items, self.p,
                                                        int(S1 is not None), int(want_rowsq))
        if need == 0:
            raise _lib.PlsrError(f"plsr_item_fused: unsupported shape n={self.n} nz={nz} k={k} ncell={ncell}")
        work = self._buf("k4work", need) if pool else torch.empty(need, dtype=torch.uint8, device=self.device)
        refd = self.dev(ref)
        vst = self._vst(items, k, pool) if want_vst else None
        k16 = (k + 15) // 16 * 16
        rowsq = torch.empty((items, k16), dtype=torch.float64, device=self.device) if want_rowsq else None
        sc = sh = None
        ready = 0
        if stats is not None:
            if "sc" in stats:
                sc, sh, ready = stats["sc"], stats["sh"], 1
            else:
                sc = stats["sc"] = torch.empty((items, ncell, self.p), dtype=torch.float64, device=self.device)
                sh = stats["sh"] = torch.empty_like(sc)
        _lib.check(self.lib.plsr_item_fused(
            _ptr(self.X), self.X.stride(0), self.p, self.n, _ptr(d_src), nz, lo, zf, ncell, _ptr(d_rows),
            items, k, _ptr(refd), _ptr(S1), _ptr(S2), _ptr(vst), self.p, _ptr(rowsq), _ptr(sc), _ptr(sh), ready,
            _ptr(work), need,
            _stream()), "plsr_item_fused")
        return vst, (rowsq[:, :k] if want_rowsq else None)

    def _vst(self, items, k, pool, ld=None, key="vst"):
        ld = self.p if ld is None else ld
        if not pool:
            return torch.empty((items, k, ld), dtype=torch.float64, device=self.device)
        return self._buf(key, items * k * ld * 8)[:items * k * ld * 8].view(torch.float64).view(items, k, ld)

    def item_beh(self, src, cell_lo, ranges, Yz, U, ref=None, S1=None, S2=None, want_vst=True, pool=False, tiled=False):
        """K4b: VS_b of behaviour PLS in two stages (plsr_item_beh), or None when the shape is not
        served.  src (items, nz), Yz (items, nz, b) z-scored within the cells, U (ncell * b, k).
        tiled: VS^T leaves tile-major, (items, p_pad / 32, k, 32) in a block of (items, k, p_pad) doubles -- the
        layout latent_batch_index streams (see plsr.h)."""
        d_src = self.dev(src, torch.int32)
        d_Yz = self.dev(Yz)
        d_U = self.dev(U)
        items, nz, b = d_Yz.shape
        k = int(d_U.shape[1])
        ncell = len(cell_lo) - 1
        lo = (ctypes.c_int32 * (ncell + 1))(*[int(x) for x in cell_lo])
        slo = (ctypes.c_int32 * ncell)(*[int(x) for x in ranges[0]])
        shi = (ctypes.c_int32 * ncell)(*[int(x) for x in ranges[1]])
        need = self.lib.plsr_item_beh_workspace_bytes(self.n, nz, b, k, lo, slo, shi, ncell, items, self.p,
                                                      int(S1 is not None))
        if not need:
            return None
        self.last_item_kernel = "beh"
        work = self._buf("k4work", need) if pool else torch.empty(need, dtype=torch.uint8, device=self.device)
        refd = self.dev(ref)
        ld = (self.p + 31) // 32 * 32 if tiled else self.p
        vst = self._vst(items, k, pool, ld) if want_vst else None
        _lib.check(self.lib.plsr_item_beh(
            _ptr(self.X), self.X.stride(0), self.p, self.n, _ptr(d_src), nz, lo, slo, shi, ncell, _ptr(d_Yz), b,
            _ptr(d_U), items, k, _ptr(refd), _ptr(S1), _ptr(S2), _ptr(vst), ld, int(tiled), _ptr(work), need,
            _stream()), "plsr_item_beh")
        return vst

    def rows_project(self, R, rowsq, U, ref=None, S1=None, S2=None, out=None):
        """K4m (plsr_rows_project): R (items, kr, p) -- the products of the UN-NORMALISED multiblock rows, from
        item_fused(want_vst=True, want_rowsq=True) with those rows as operator -- becomes VS^T in place (or goes to
        `out` (items, k, p), R left as it is), VS_b = (U^T D_b^-1) R_b with D_b = sqrt(rowsq_b); the shifted moment
        sums are added to S1 / S2.  Returns False (and does nothing) when the shape is not served."""
        items, kr, p = R.shape
        d_U = self.dev(U)
        k = int(d_U.shape[1])
        need = self.lib.plsr_rows_project_workspace_bytes(kr, k, items, p, int(S1 is not None))
        if not need or k != kr or int(d_U.shape[0]) != kr:
            return False
        work = self._buf("k4mwork", need)
        _lib.check(self.lib.plsr_rows_project(_ptr(R), R.stride(1), p, items, kr, _ptr(rowsq), rowsq.stride(0), _ptr(d_U),
                                              k, _ptr(self.dev(ref)), _ptr(S1), _ptr(S2), _ptr(out), _ptr(work), need,
                                              _stream()), "plsr_rows_project")
        return True

    @staticmethod
    def _batch_bounds(R, step):
        """Batches of `step` resamples with a short first and a short last one: the device starts
        after a quarter batch's worth of host preparation, and what stays exposed at the end -- the
        last batch's device time and its host post-processing -- is a quarter batch, too."""
        q = max(4, step // 4 // 4 * 4)
        if R <= 2 * step or q >= step:
            return [(lo, min(R, lo + step)) for lo in range(0, R, step)]
        out, lo = [(0, q)], q
        while R - lo > step + q:
            out.append((lo, lo + step))
            lo += step
        if R - lo > q:
            out.append((lo, R - q))
            lo = R - q
        out.append((lo, R))
        return out

    def boot_items(self, src, cell_lo, cell_z, k, ops_fn, ref=None, raw_rows_fn=None, latent_rows=None,
                   on_batch=None, project_on=None, beh=None, after_enqueue=None, need_nsq=True, cells_fn=None,
                   latent_index=None, own_rows=False):
        """Bootstrap phase in which every resample has its own gathered /
        z-scored matrix (behaviour and multiblock PLS).

        src (R, n') int32 + cell_lo / cell_z : what K3 builds per resample;
        ops_fn(lo, hi, rownorm) -> (hi-lo, k, n') operator rows with
            VS_b = ops_b @ Z_b; rownorm is None, or -- when raw_rows_fn is
            given -- the (hi-lo, m) norms over all voxels of the rows
            raw_rows_fn(lo, hi) @ Z_b (the multiblock row normalisation);
        project_on: (m, k) matrix U.  With raw_rows_fn, the operator rows are then
            formed on the device, ops_b = U^T diag(1 / rownorm_b) raw_b, and ops_fn
            is not called -- the host never waits for the norms;
        latent_rows: number of leading rows of X used for X @ VS_b (default n).
        on_batch(lo, hi, Zt_host, nsq_host, form): optional consumer of every batch's
            latent scores (NumPy, (hi-lo, k, width) and (hi-lo, k)); it is called one
            batch late, while the device already works on the next batch, so the
            host's per-resample post-processing hides behind the kernels.  form: "rows" -- width n, column i =
            row i of X; "index" -- width m, column i = row latent_index[b, i]; "index+own" -- width m + t, the
            last t columns the products of VS_b with the item's own raw task rows (own_rows).
        cells_fn(lo, hi) -> (cell description of the batch's items as engine.split_rows takes it, Y): with
            project_on, the un-normalised rows come from the two-stage kernel (plsr_split_rows) and the
            projection from the stream over them (plsr_rows_project); raw_rows_fn / ops_fn are then not
            called.  Falls back to them when a shape is not served.
        need_nsq=False: the caller does not use the column norms (behaviour PLS: its per-cell
            z-score of the latent scores is scale invariant); the latent kernel then skips them
            and nsq comes back as NaN.
        latent_index (R, m) int: the rows of X[:n] whose latent scores the caller reads per resample (its
            sample, `X_new @ V_hat`): on_batch's scores are then (.., k, m), column i = row
            latent_index[b, i] (computed once per DIFFERENT row of the sample, latent_batch_index).
        own_rows: with cells_fn and latent_index -- the caller also wants VS_b times the raw TASK rows of the item
            (the rows of the cells description with row_cell < 0, in row_sub order): the multiblock's Tdistrib,
            `cell means of smeanmat(X_new_T) @ V_hat` = raw task rows @ V_hat.  Batches the two-stage path serves
            come back as "index+own"; the others as "rows" (the caller then forms both from the full scores).
        Returns dict(S1, S2 (p x k shifted moment sums),
        nsq (R, k) = column norms^2 of VS_b)."""
        index_is_src = latent_index is src
        src = np.ascontiguousarray(src, dtype=np.int32)
        R, nz = src.shape
        n = self.n if latent_rows is None else int(latent_rows)
        refd = self.dev(ref)
        S12 = torch.zeros((2, self.p, k), dtype=torch.float64, device=self.device)
        S1, S2 = S12[0], S12[1]
        if latent_index is not None:
            latent_index = src if index_is_src else np.ascontiguousarray(latent_index, dtype=np.int32)
            if latent_index.shape[0] != R or latent_index.ndim != 2:
                raise ValueError("latent_index must hold one row per resample")
        nsq = torch.empty((R, k), dtype=torch.float64, device=self.device)
        if not need_nsq:
            nsq.fill_(float("nan"))
        ncell = len(cell_z)
        per_item = (2 * ncell + k + 4) * self.p * 8 + 2 * k * nz * 8
        step = int(max(1, min(R, (self.work_limit // 2) // per_item)))
        pending = []

        def deliver(job):
            blo, bhi, ev, Zb, form = job
            if self._d2h is None:
                self._d2h = _side_stream(self.device, "d2h")
            # into page-locked buffers (torch's caching host allocator recycles them): a copy into
            # fresh pageable memory makes the runtime lock and unlock those pages for the transfer,
            # i.e. update the GPU's page tables while the next batch's kernels run
            with torch.cuda.stream(self._d2h):            # waits for that batch only, not for the stream's tail
                self._d2h.wait_event(ev)
                zt_h = torch.empty(Zb.shape, dtype=Zb.dtype, pin_memory=True)
                nsq_h = torch.empty(nsq[blo:bhi].shape, dtype=nsq.dtype, pin_memory=True)
                zt_h.copy_(Zb, non_blocking=True)
                nsq_h.copy_(nsq[blo:bhi], non_blocking=True)
                done = torch.cuda.Event()
                done.record(self._d2h)
            done.synchronize()
            on_batch(blo, bhi, zt_h.numpy(), nsq_h.numpy(), form)

        # the cells of a bootstrap sample read fixed ranges of source rows: aggregated-operator
        # kernel (K4a) when the shape allows; it leaves the column norms to the latent kernel
        ranges = self.source_ranges(src, cell_lo)
        for lo, hi in self._batch_bounds(R, step):
            cnt = hi - lo
            d_src = self.dev(src[lo:hi], torch.int32)
            use_agg = self._agg_bytes(nz, k, cell_lo, cell_z, ranges, cnt, True, False) > 0
            rownorm = None
            vst = None
            own = None
            most = tiled = None
            if latent_index is not None:
                most = self.distinct_rows(latent_index[lo:hi]) if n <= 128 else 0
            if cells_fn is not None and project_on is not None and np.shape(project_on)[0] == k == np.shape(project_on)[1] \
                    and self.lib.plsr_rows_project_workspace_bytes(k, k, cnt, self.p, 1) > 0:
                cells, Ycells = cells_fn(lo, hi)
                got = self.split_rows(cells, Ycells, pool=True) if cells is not None else None
                if got is not None:
                    vst, rsq = got
                    # with own_rows the raw rows stay (the latent kernel reads the task rows among them) and VS^T gets
                    # a block of its own
                    trows = [i for _, i in sorted((int(sb), i) for i, (c, sb) in
                                                  enumerate(zip(cells["row_cell"], cells["row_sub"])) if c < 0)]
                    if own_rows and latent_index is not None and trows and len(trows) <= 16 and \
                            self.index_served(n, k, cnt, latent_index.shape[1], most, len(trows)):
                        own = (vst, trows)
                        vst = self._vst(cnt, k, True, key="vsout")
                    if not self.rows_project(own[0] if own else vst, rsq, project_on, ref=refd, S1=S1, S2=S2,
                                             out=vst if own else None):
                        raise _lib.PlsrError("plsr_rows_project declined a shape its workspace query accepted")
                    self.last_item_kernel = "rows+project"
                    use_agg = True                   # (column norms from the latent kernel)
            if vst is not None:
                pass
            elif raw_rows_fn is not None:
                # two-phase row normalisation of the multiblock (class_functions.py:503-505):
                # norms over all voxels of the un-normalised rows, K4a / K4f
                raw = np.ascontiguousarray(raw_rows_fn(lo, hi), dtype=np.float64)
                d_raw = self.dev(raw)
                stats = {}                       # both passes run on the same items: statistics once
                m_raw = int(d_raw.shape[1])
                stream_second = (project_on is not None and use_agg and np.shape(project_on) == (m_raw, k) and m_raw == k
                                 and self.lib.plsr_rows_project_workspace_bytes(m_raw, k, cnt, self.p, 1) > 0)
                if stream_second:
                    # ... which also stores the rows' products; the projection on U is then a stream over them
                    # (K4m, plsr_rows_project: 0.6 instead of 3 GFLOP per item), in place
                    vst, rsq = self.item_fused(d_src, cell_lo, cell_z, d_raw, want_vst=True, want_rowsq=True,
                                               src_ranges=ranges, pool=True)
                    if not self.rows_project(vst, rsq, project_on, ref=refd, S1=S1, S2=S2):
                        raise _lib.PlsrError("plsr_rows_project declined a shape its workspace query accepted")
                    self.last_item_kernel += "+rows"
                else:
                    _, rsq = self.item_fused(d_src, cell_lo, cell_z, d_raw, want_rowsq=True, stats=stats,
                                             src_ranges=ranges, pool=True)
                    if project_on is None:
                        rownorm = np.sqrt(rsq.cpu().numpy())
            if vst is not None:
                ops = None
            elif raw_rows_fn is not None and project_on is not None:
                d_U = self.dev(project_on)
                m = int(d_raw.shape[1])
                ops = torch.empty((cnt, k, nz), dtype=torch.float64, device=self.device)
                _lib.check(self.lib.plsr_scale_project_rows(_ptr(d_raw), _ptr(rsq), rsq.stride(0), _ptr(d_U), cnt,
                                                            m, nz, k, _ptr(ops), _stream()),
                           "plsr_scale_project_rows")
            elif beh is not None and self.lib.plsr_item_beh_workspace_bytes(
                    self.n, nz, int(np.shape(beh[1])[0]) // ncell, k,
                    (ctypes.c_int32 * (ncell + 1))(*[int(x) for x in cell_lo]),
                    (ctypes.c_int32 * ncell)(*[int(x) for x in ranges[0]]),
                    (ctypes.c_int32 * ncell)(*[int(x) for x in ranges[1]]), ncell, cnt, self.p, 1):
                ops = None                       # (the two-stage kernel serves: no dense operator rows)
            else:
                ops = np.ascontiguousarray(ops_fn(lo, hi, rownorm), dtype=np.float64)  # (cnt, k, nz)
            if beh is not None and raw_rows_fn is None:
                # behaviour PLS: the two-stage kernel takes the z-scored behaviour rows and U themselves; its VS^T
                # leaves tile-major when the latent kernel that streams that layout follows
                tiled = latent_index is not None and self.index_served(n, k, cnt, latent_index.shape[1], most)
                vst = self.item_beh(d_src, cell_lo, ranges, beh[0](lo, hi), beh[1], ref=refd, S1=S1, S2=S2, pool=True,
                                    tiled=tiled)
                tiled = tiled and vst is not None
                use_agg = use_agg or vst is not None        # (column norms from the latent kernel)
            if vst is None:
                vst, rsq = self.item_fused(d_src, cell_lo, cell_z, ops, ref=refd, S1=S1, S2=S2, want_vst=True,
                                           want_rowsq=not use_agg, stats=stats if raw_rows_fn is not None else None,
                                           src_ranges=ranges if use_agg else None, pool=True)
            if not use_agg:
                nsq[lo:hi] = rsq
            nsq_b = nsq[lo:hi] if use_agg and need_nsq else None
            if latent_index is not None and (own is not None or not own_rows):
                m_ix = latent_index.shape[1]
                form = "index+own" if own else "index"
                Zb = torch.empty((cnt, k, m_ix + (len(own[1]) if own else 0)), dtype=torch.float64, device=self.device)
                d_li = d_src if index_is_src else self.dev(latent_index[lo:hi], torch.int32)
                self.latent_batch_index(vst, n, latent_index[lo:hi], d_li, Zb, nsq_b, tiled=bool(tiled), most=most, own=own)
            else:
                form = "rows"
                Zb = torch.empty((cnt, k, n), dtype=torch.float64, device=self.device)
                self.latent_batch(vst, n, Zb, nsq_b)
            if on_batch is not None:
                ev = torch.cuda.Event()
                ev.record()
                pending.append((lo, hi, ev, Zb, form))
                # Two batches stay enqueued behind the one the device runs: the host consumes batch
                # i - 2 while batch i - 1 runs and batch i waits.  With one batch of slack a late host
                # (a slow on_batch) left the device idle for a fraction of a millisecond now and then
                # -- and a device that has idled drops its clocks: the next kernel then ran 20-40 ms
                # instead of 6 (the "sporadic slow launch" of the kernel trace).
                while len(pending) > 2:
                    deliver(pending.pop(0))
        # whatever only needs the moment sums (std_errs / boot_ratios and their download) is enqueued
        # now, behind the last batch, and proceeds while the host consumes the batches still pending
        after = after_enqueue(S1, S2) if after_enqueue is not None else None
        while pending:
            deliver(pending.pop(0))
        return {"S1": S1, "S2": S2, "S12": S12, "nsq": nsq, "R": R, "after": after}

    def eigh(self, G, off, k, init=None, relative=False):
        """Eigen-decomposition of the k x k diagonal block at `off` of every
        matrix in G (S, mm, mm): (evals (S,k) descending, evecs (S,k,k)).
        init (S, k, k): basis the rotations are applied to (evecs = init @ J);
        relative: refinement pass on a graded, nearly diagonal Gram (see plsr.h)."""
        S, mm, _ = G.shape
        ev = torch.empty((S, k), dtype=torch.float64, device=self.device)
        vec = torch.empty((S, k, k), dtype=torch.float64, device=self.device)
        _lib.check(self.lib.plsr_eigh_batch(_ptr(G), mm * mm, mm, off, k, S, _ptr(ev), _ptr(vec),
                                            _ptr(init), int(relative), _stream()), "plsr_eigh_batch")
        return ev, vec

    def rotate_rows(self, U, rows, off=0):
        """rows (S, m, n) device tensor with the block of k rows at `off` expressed in the
        basis U (S, k, k):  out[s, off + j] = sum_i U[s, i, j] rows[s, off + i]."""
        S, m, n = rows.shape
        k = U.shape[-1]
        out = torch.empty_like(rows)
        _lib.check(self.lib.plsr_rotate_rows(_ptr(U), _ptr(rows), _ptr(out), S, m, n, off, k, _stream()),
                   "plsr_rotate_rows")
        return out

    SVD_PASSES = 3

    def thin_svd_device(self, rows, null_tol=None):
        """Thin SVD of M = rows @ X (k x p), everything on the device, no host synchronisation.
        Returns dict(U (k, k), s (k,) descending with null values deflated to 0, VSt (k, p) =
        (V s)^T, Vt (k, p) = V^T, UtR (k, n) = U^T rows) of device tensors.

        Pass 0: Gram of the rows -> Jacobi.  That alone squares the condition number (a singular
        value 1e-4 of the largest comes out with 1e-8 relative error).  Passes 1..: the rows are
        expressed in the basis found so far (they are then nearly orthogonal with norms close to the
        singular values), their Gram is formed again -- entry (i, j) now carries an error of
        eps s_i s_j instead of eps s_1^2 -- and Jacobi in relative mode continues from the
        accumulated basis.  Two passes reach LAPACK's accuracy (each pass squares the remaining
        error); a third is run as a safeguard.  Measured against an 80-bit one-sided Jacobi on
        graded spectra (s_k / s_1 = 1e-7, k = 48): 2e-4 relative after pass 0, 2e-12 after pass 1,
        where LAPACK's dgesdd itself is off by 4e-11."""
        d_rows = (rows if torch.is_tensor(rows) else self.dev(np.ascontiguousarray(rows, dtype=np.float64)))[None]
        k, n = int(d_rows.shape[1]), int(d_rows.shape[2])
        cur, U, ev = d_rows, None, None
        for it in range(self.SVD_PASSES):
            G = self.gram_phase(cur)
            ev, U = self.eigh(G, 0, k, init=U, relative=it > 0)
            cur = self.rotate_rows(U, d_rows)              # U^T rows: next pass's rows / back-projection operator
        s = torch.empty(k, dtype=torch.float64, device=self.device)
        ops = torch.empty((2 * k, n), dtype=torch.float64, device=self.device)
        abs_tol, rel_tol = (1e-12, 4.0 * k * np.finfo(float).eps) if null_tol is None else (0.0, float(null_tol))
        _lib.check(self.lib.plsr_svd_finish(_ptr(ev), _ptr(cur), k, n, abs_tol, rel_tol, _ptr(s), _ptr(ops),
                                            _stream()), "plsr_svd_finish")
        both = self.apply_operator(ops)                    # (2k, p): (V s)^T and V^T in one pass over X
        return {"U": U[0], "s": s, "VSt": both[:k], "Vt": both[k:], "UtR": cur[0]}

    @staticmethod
    def null_threshold(k, smax):
        """Singular values at or below this are returned as exactly 0 (vectors 0): the reference's
        absolute 1e-12 (bootstrap_permutation.py:295), and whatever lies below the resolution of
        the reference's own LAPACK SVD (a few k eps s_max: the null latent variables of a
        rank-deficient centring come out of dgesdd as noise of that size)."""
        return max(1e-12, 4.0 * k * np.finfo(float).eps * smax)

    def thin_svd(self, rows, null_tol=None):
        """Thin SVD of M = rows @ X (k x p) without forming M on the host (class_functions.py:98-123):
        (U, s, V) as NumPy arrays.  Null singular values are deflated to 0 with zero vectors
        (null_threshold; `null_tol` overrides it with a threshold relative to s_max)."""
        d = self.thin_svd_device(np.asarray(rows, dtype=float), null_tol)
        U, s, Vt = self.fetch_async([d["U"], d["s"], d["Vt"]]).get()
        return U, s, Vt.T
