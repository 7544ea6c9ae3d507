// extern "C" surface of libplsr_hip.so (declared in include/plsr.h).
// Host code here only validates shapes, carves the caller's workspace and
// enqueues kernels on the caller's stream; it never allocates or synchronises.
#include "../../include/plsr.h"
#include "plsr_project.hip.h"

#include <algorithm>
#include <numeric>
#include <vector>

using namespace plsr;

static thread_local int g_last_hip = 0;

// optional per-launch timing of the projection kernel (plsr_timing_*)
namespace {
struct TimedLaunch {
  hipEvent_t a, b;
  int kind;
};
bool g_timing = false;
std::vector<TimedLaunch> g_timed;
}  // namespace

static inline int check_launch() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    g_last_hip = (int)e;
    return PLSR_ELAUNCH;
  }
  return PLSR_OK;
}

extern "C" int plsr_abi_version(void) { return 1; }
extern "C" int plsr_last_hip_error(void) { return g_last_hip; }

extern "C" const char *plsr_strerror(int code) {
  switch (code) {
    case PLSR_OK: return "ok";
    case PLSR_EINVAL: return "invalid argument";
    case PLSR_EUNSUPPORTED: return "shape not supported by the gfx950 kernels";
    case PLSR_EWORKSPACE: return "workspace too small";
    case PLSR_ELAUNCH: return "HIP launch failed";
    default: return "unknown error";
  }
}

extern "C" int plsr_layout_init(int32_t n, int32_t k, int32_t R, plsr_layout_t *out) {
  if (!out || n <= 0 || k <= 0 || R <= 0) return PLSR_EINVAL;
  int kp = k;
  int period = k / std::gcd(4, k);
  if (period > MAX_PERIOD) {
    kp = (k + 3) / 4 * 4;
    period = kp / 4;
  }
  if (period > MAX_PERIOD) return PLSR_EUNSUPPORTED;
  out->n = n;
  out->k = k;
  out->R = R;
  out->nk = (n + 3) / 4;
  out->kp = kp;
  out->period = period;
  out->Rp = (R + 3) / 4 * 4;
  const int64_t nquads = (int64_t)kp * (out->Rp / 4);
  out->ntiles = (int32_t)((nquads + 3) / 4);
  out->frag_elems = (int64_t)out->ntiles * out->nk * 64;
  // the X tile plus the transpose patches must fit the 160 KiB LDS of a CU
  if (project_lds_bytes(out->nk, period, true) > 160 * 1024) return PLSR_EUNSUPPORTED;
  return PLSR_OK;
}

static inline int64_t lay_nquads(const plsr_layout_t *l) { return (int64_t)l->kp * (l->Rp / 4); }

static int launch_ops(const int32_t *d_inds, const double *d_M, const double *d_cols,
                      const plsr_layout_t *lay, double *d_frag, void *stream) {
  if (!lay || !d_frag) return PLSR_EINVAL;
  OpsArgs a;
  a.inds = d_inds;
  a.M = d_M;
  a.cols = d_cols;
  a.frag = d_frag;
  a.n = lay->n;
  a.nk = lay->nk;
  a.k = lay->k;
  a.kp = lay->kp;
  a.R = lay->R;
  a.nquads = (int32_t)lay_nquads(lay);
  a.ntiles = lay->ntiles;
  const int64_t total = lay->frag_elems;
  dim3 grid((unsigned)((total + 255) / 256));
  if (d_cols)
    hipLaunchKernelGGL(ops_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(ops_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, a);
  return check_launch();
}

extern "C" int plsr_ops_from_indices(const int32_t *d_inds, const double *d_M,
                                     const plsr_layout_t *lay, double *d_frag, void *stream) {
  if (!d_inds || !d_M) return PLSR_EINVAL;
  return launch_ops(d_inds, d_M, nullptr, lay, d_frag, stream);
}

extern "C" int plsr_ops_pack(const double *d_cols, const plsr_layout_t *lay, double *d_frag,
                             void *stream) {
  if (!d_cols) return PLSR_EINVAL;
  return launch_ops(nullptr, nullptr, d_cols, lay, d_frag, stream);
}

// ---------------------------------------------------------------------------
// workspace carving
// ---------------------------------------------------------------------------
namespace {
constexpr int SLAB_CHUNK = 64;

struct Work {
  double *norm_part;   // [nvt][C]
  double *T_part;      // [nvt][C][k2]
  double *lvl2;        // [nchunk][C*max(1,k2)]
  int64_t nvt, C;
  int nchunk;
  size_t bytes;
};

Work carve(const plsr_layout_t *lay, int64_t p, int32_t k2, void *base) {
  Work w;
  w.nvt = (p + TV - 1) / TV;
  w.C = (int64_t)lay->ntiles * 16;
  w.nchunk = (int)((w.nvt + SLAB_CHUNK - 1) / SLAB_CHUNK);
  size_t off = 0;
  auto take = [&](size_t elems) {
    double *ptr = base ? (double *)((char *)base + off) : nullptr;
    off += (elems * sizeof(double) + 255) / 256 * 256;
    return ptr;
  };
  w.norm_part = take((size_t)w.nvt * w.C);
  w.T_part = k2 > 0 ? take((size_t)w.nvt * w.C * k2) : nullptr;
  w.lvl2 = take((size_t)w.nchunk * w.C * std::max(1, (int)k2));
  w.bytes = off;
  return w;
}

int reduce_slabs(const double *slabs, const Work &w, int width, const plsr_layout_t *lay,
                 double *d_out, hipStream_t st) {
  const int64_t E = w.C * width;
  dim3 g1((unsigned)((E + 255) / 256), (unsigned)w.nchunk);
  hipLaunchKernelGGL(slab_sum_kernel, g1, dim3(256), 0, st, slabs, w.lvl2, E, (int)w.nvt,
                     SLAB_CHUNK);
  dim3 g2((unsigned)((E + 255) / 256));
  hipLaunchKernelGGL(slab_final_kernel, g2, dim3(256), 0, st, (const double *)w.lvl2, d_out, w.C,
                     width, w.nchunk, lay->kp, lay->k, lay->R, (int)lay_nquads(lay));
  return check_launch();
}

template <int MODE>
int launch_project(const ProjectArgs &a, int period, int64_t nvt, hipStream_t st) {
  const size_t lds = project_lds_bytes(a.nk, period, MODE != 0);
  dim3 grid((unsigned)nvt), block(256);
#define PLSR_CASE(P)                                                                          \
  case P: {                                                                                   \
    auto kern = project_kernel<P, MODE>;                                                      \
    if (lds > 64 * 1024) {                                                                    \
      hipError_t e = hipFuncSetAttribute((const void *)kern,                                  \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      if (e != hipSuccess) {                                                                  \
        g_last_hip = (int)e;                                                                  \
        return PLSR_ELAUNCH;                                                                  \
      }                                                                                       \
    }                                                                                         \
    TimedLaunch tl{};                                                                         \
    if (g_timing) {                                                                           \
      (void)hipEventCreate(&tl.a);                                                                  \
      (void)hipEventCreate(&tl.b);                                                                  \
      tl.kind = MODE;                                                                 \
      (void)hipEventRecord(tl.a, st);                                                               \
    }                                                                                         \
    hipLaunchKernelGGL(kern, grid, block, lds, st, a);                                        \
    if (g_timing) {                                                                           \
      (void)hipEventRecord(tl.b, st);                                                               \
      g_timed.push_back(tl);                                                                  \
    }                                                                                         \
    break;                                                                                    \
  }
  switch (period) {
    PLSR_CASE(1)
    PLSR_CASE(2)
    PLSR_CASE(3)
    PLSR_CASE(4)
    PLSR_CASE(5)
    PLSR_CASE(6)
    default: return PLSR_EUNSUPPORTED;
  }
#undef PLSR_CASE
  return check_launch();
}
}  // namespace

extern "C" size_t plsr_batch_workspace_bytes(const plsr_layout_t *lay, int64_t p, int32_t k2) {
  if (!lay || p <= 0 || k2 < 0) return 0;
  return carve(lay, p, k2, nullptr).bytes;
}

static int fill_common(ProjectArgs &a, const double *d_X, int64_t ldx, int64_t p,
                       const double *d_frag, const plsr_layout_t *lay) {
  if (!d_X || !d_frag || !lay || p <= 0 || ldx < p) return PLSR_EINVAL;
  if (lay->period < 1 || lay->period > MAX_PERIOD) return PLSR_EUNSUPPORTED;
  a = ProjectArgs{};
  a.X = d_X;
  a.ldx = ldx;
  a.p = p;
  a.n = lay->n;
  a.nk = lay->nk;
  a.frag = d_frag;
  a.ntiles = lay->ntiles;
  a.k = lay->k;
  a.kp = lay->kp;
  a.R = lay->R;
  a.nquads = (int32_t)lay_nquads(lay);
  return PLSR_OK;
}

extern "C" int plsr_perm_batch(const double *d_X, int64_t ldx, int64_t p, const double *d_frag,
                               const plsr_layout_t *lay, double *d_ssq, void *d_work,
                               size_t work_bytes, void *stream) {
  ProjectArgs a;
  int rc = fill_common(a, d_X, ldx, p, d_frag, lay);
  if (rc) return rc;
  if (!d_ssq || !d_work) return PLSR_EINVAL;
  Work w = carve(lay, p, 0, d_work);
  if (w.bytes > work_bytes) return PLSR_EWORKSPACE;
  a.norm_part = w.norm_part;
  hipStream_t st = (hipStream_t)stream;
  rc = launch_project<0>(a, lay->period, w.nvt, st);
  if (rc) return rc;
  return reduce_slabs(w.norm_part, w, 1, lay, d_ssq, st);
}

extern "C" int plsr_boot_batch(const double *d_X, int64_t ldx, int64_t p, const double *d_frag,
                               const plsr_layout_t *lay, const double *d_ref, const double *d_Xm,
                               int64_t ldxm, int32_t k2, double *d_S1, double *d_S2,
                               double *d_ssq, double *d_T, double *d_vs_dump, void *d_work,
                               size_t work_bytes, void *stream) {
  ProjectArgs a;
  int rc = fill_common(a, d_X, ldx, p, d_frag, lay);
  if (rc) return rc;
  if (!d_S1 || !d_S2 || !d_ssq || !d_work) return PLSR_EINVAL;
  if (k2 < 0 || k2 > 16) return PLSR_EUNSUPPORTED;
  if (k2 > 0 && (!d_Xm || !d_T || ldxm < p)) return PLSR_EINVAL;
  Work w = carve(lay, p, k2, d_work);
  if (w.bytes > work_bytes) return PLSR_EWORKSPACE;
  a.norm_part = w.norm_part;
  a.T_part = w.T_part;
  a.ref = d_ref;
  a.Xm = k2 > 0 ? d_Xm : nullptr;
  a.ldxm = ldxm;
  a.k2 = k2;
  a.S1 = d_S1;
  a.S2 = d_S2;
  a.vs_dump = d_vs_dump;
  hipStream_t st = (hipStream_t)stream;
  rc = d_vs_dump ? launch_project<2>(a, lay->period, w.nvt, st)
                 : launch_project<1>(a, lay->period, w.nvt, st);
  if (rc) return rc;
  rc = reduce_slabs(w.norm_part, w, 1, lay, d_ssq, st);
  if (rc) return rc;
  if (k2 > 0) rc = reduce_slabs(w.T_part, w, k2, lay, d_T, st);
  return rc;
}

extern "C" int plsr_boot_finalize(const double *d_S1, const double *d_S2, const double *d_num,
                                  int64_t count, int32_t R, double *d_std, double *d_ratio,
                                  void *stream) {
  if (!d_S1 || !d_S2 || !d_std || count <= 0 || R <= 0) return PLSR_EINVAL;
  if (d_ratio && !d_num) return PLSR_EINVAL;
  dim3 grid((unsigned)((count + 255) / 256));
  hipLaunchKernelGGL(boot_finalize_kernel, grid, dim3(256), 0, (hipStream_t)stream, d_S1, d_S2,
                     d_num, count, 1.0 / (double)R, d_std, d_ratio);
  return check_launch();
}

extern "C" int plsr_timing_enable(int on) {
  g_timing = on != 0;
  return PLSR_OK;
}

extern "C" int plsr_timing_collect(double *ms_out, int32_t *kind_out, int32_t max) {
  int n = 0;
  for (auto &tl : g_timed) {
    (void)hipEventSynchronize(tl.b);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, tl.a, tl.b);
    if (n < max && ms_out) {
      ms_out[n] = ms;
      if (kind_out) kind_out[n] = tl.kind;
      ++n;
    }
    (void)hipEventDestroy(tl.a);
    (void)hipEventDestroy(tl.b);
  }
  g_timed.clear();
  return n;
}
