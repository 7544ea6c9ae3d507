// extern "C" surface of libplsr_hip.so (declared in include/plsr.h).
// Host code here only validates shapes, carves the caller's workspace and
// enqueues kernels on the caller's stream; it never allocates or synchronises.
#include "../../include/plsr.h"
#include "plsr_project.hip.h"

#include <algorithm>
#include <cstdlib>
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

// optional side stream for the reduction tail of plsr_boot_batch
static thread_local hipStream_t g_tail = nullptr;
static thread_local hipEvent_t g_chain_ev = nullptr;

// `after` waits for everything enqueued so far on `before`
static int chain(hipStream_t before, hipStream_t after) {
  if (!g_chain_ev && hipEventCreateWithFlags(&g_chain_ev, hipEventDisableTiming) != hipSuccess)
    return PLSR_ELAUNCH;
  hipError_t e = hipEventRecord(g_chain_ev, before);
  if (e == hipSuccess) e = hipStreamWaitEvent(after, g_chain_ev, 0);
  if (e != hipSuccess) {
    g_last_hip = (int)e;
    return PLSR_ELAUNCH;
  }
  return PLSR_OK;
}

static inline int check_launch() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    g_last_hip = (int)e;
    return PLSR_ELAUNCH;
  }
  return PLSR_OK;
}

extern "C" int plsr_abi_version(void) { return 2; }
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
  if ((n + 3) / 4 >= REG_NK_MIN && (n + 3) / 4 <= REG_NK_MAX) {
    // n <= 64: the register-resident kernels (K1r / K1br) take the batch in
    // LV-major order -- tile t holds 16 consecutive resamples of latent variable
    // t / (Rp/16); period 0 marks this layout
    out->n = n;
    out->k = k;
    out->R = R;
    out->nk = (n + 3) / 4;
    out->kp = k;
    out->period = 0;
    out->Rp = (R + 15) / 16 * 16;
    out->ntiles = k * (out->Rp / 16);
    out->frag_elems = (int64_t)out->ntiles * out->nk * 64 + 4 * 64;
    return PLSR_OK;
  }
  int kp = k;
  int period = k / std::gcd(4, k);
  if (period > MAX_PERIOD) {
    kp = (k + 3) / 4 * 4;
    period = kp / 4;
  }
  out->n = n;
  out->k = k;
  out->R = R;
  out->nk = (n + 3) / 4;
  out->kp = kp;
  out->period = period;
  out->Rp = (R + 3) / 4 * 4;
  const int64_t nquads = (int64_t)kp * (out->Rp / 4);
  out->ntiles = (int32_t)((nquads + 3) / 4);
  // + four k-steps of padding: the kernel's fragment prefetch runs 4 steps ahead
  out->frag_elems = (int64_t)out->ntiles * out->nk * 64 + 4 * 64;
  // the X tile plus the transpose patches must fit the 160 KiB LDS of a CU
  if (project_lds_bytes(out->nk, std::min(period, MAX_PERIOD), true, 4, (int)out->kp) > 160 * 1024) return PLSR_EUNSUPPORTED;
  return PLSR_OK;
}

static inline int64_t lay_nquads(const plsr_layout_t *l) { return (int64_t)l->kp * (l->Rp / 4); }
// tiles per latent variable in the LV-major layout (0 = quad layout)
static inline int lay_tpl(const plsr_layout_t *l) { return l->period == 0 ? l->Rp / 16 : 0; }

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
  a.tpl = lay_tpl(lay);
  const int64_t total = (int64_t)lay->ntiles * lay->nk * 64;
  dim3 grid((unsigned)((total + 255) / 256));
  const size_t lds_sel = (size_t)16 * lay->n * (sizeof(double) + sizeof(int32_t));
  if (d_cols)
    hipLaunchKernelGGL(ops_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a);
  else if (lds_sel <= 64 * 1024)
    hipLaunchKernelGGL(ops_select_kernel, dim3((unsigned)lay->ntiles), dim3(256), lds_sel, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(ops_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, a);
  return check_launch();
}

extern "C" int plsr_ops_from_indices(const int32_t *d_inds, const double *d_M,
                                     const plsr_layout_t *lay, double *d_frag, void *stream) {
  if (!d_inds || !d_M) return PLSR_EINVAL;
  return launch_ops(d_inds, d_M, nullptr, lay, d_frag, stream);
}

extern "C" int plsr_ops_from_behaviour(const double *d_Yz, int32_t nb, const double *d_U,
                                       const int32_t *d_rowcell, const plsr_layout_t *lay, double *d_frag,
                                       void *stream) {
  if (!d_Yz || !d_U || !d_rowcell || !lay || !d_frag || nb <= 0) return PLSR_EINVAL;
  OpsBehArgs b;
  b.Yz = d_Yz;
  b.U = d_U;
  b.rowcell = d_rowcell;
  b.nb = nb;
  b.o = OpsArgs{};
  b.o.frag = d_frag;
  b.o.n = lay->n;
  b.o.nk = lay->nk;
  b.o.k = lay->k;
  b.o.kp = lay->kp;
  b.o.R = lay->R;
  b.o.nquads = (int32_t)lay_nquads(lay);
  b.o.ntiles = lay->ntiles;
  b.o.tpl = lay_tpl(lay);
  const int64_t total = (int64_t)lay->ntiles * lay->nk * 64;
  hipLaunchKernelGGL(ops_behaviour_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, b);
  return check_launch();
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
  double *mom_part;    // [2][nsplit][p][k]   (boot; LV-major layout: [1 + nsplit][p][k])
  double *opsum;       // [4 nk][k]           (boot, LV-major layout)
  double *sink;        // [64] store target of idle lanes (register-resident kernels)
  int nsplit;          // column splits of the batch (grid.y)
  double *norm_part;   // [nvt][C]
  double *T_part;      // [nvt][C][k2]
  double *lvl2;        // [nchunk][C*max(1,k2)]
  int64_t nvt, C;
  int nchunk;
  size_t bytes;
};

// column splits: pick the grid.y in 1..4 that wastes least in the last round of
// workgroups (256 CUs x 2 resident workgroups), without splitting below one
// wave x period group per workgroup
int pick_split(const plsr_layout_t *lay, int64_t nvt, int nw) {
  if (lay->period == 0) {
    // LV-major layout (K1br): splits per latent variable, for about ten rounds of
    // the 2048 resident waves (one wave per workgroup, grid.y = k * splits); measured
    // at config 2: 5 / 10 / 20 / 30 rounds -> 2.71 / 2.69 / 2.72 / 2.81 ms, and fewer
    // runs re-read X less often
    // ... and runs of at least eight tiles: a run loads its 4 nk X fragments once, and with four-tile runs
    // (125 resamples per launch, a rank's share of config 2 on eight GPUs) the launch took 0.53 ms where
    // 0.32 would be its share of the full launch
    // (where the grid still has eight rounds of waves with them; small p keeps runs of four)
    const int tpl = lay->Rp / 16;
    const int64_t want = (10 * 2048 + nvt * lay->k - 1) / (nvt * lay->k);
    const int64_t m4 = std::max<int64_t>(1, std::min<int64_t>(want, std::max(1, tpl / 4)));
    const int64_t m8 = std::max<int64_t>(1, std::min<int64_t>(want, std::max(1, tpl / 8)));
    return (int)(nvt * lay->k * m8 >= 8 * 2048 ? m8 : m4);
  }
  const int groups = (lay->ntiles + nw * lay->period - 1) / (nw * lay->period);
  int best = 1;
  double best_eff = 0.0;
  for (int c = 1; c <= 4 && c <= groups; ++c) {
    const double rounds = (double)nvt * c / 512.0;
    const double eff = rounds / std::ceil(rounds);
    if (eff > best_eff + 1e-9) {
      best_eff = eff;
      best = c;
    }
  }
  return best;
}

Work carve(const plsr_layout_t *lay, int64_t p, int32_t k2, void *base, bool boot) {
  Work w;
  w.nvt = (p + TV - 1) / TV;
  w.C = (int64_t)lay->ntiles * 16;
  // (waves per workgroup of the LDS-fed kernel: the permutation launch runs the period-1 instance)
  const int nw = lay->period == 0 ? WAVES
                 : lds_fed_waves(lay->nk, boot ? lay->period : 1, boot, boot ? (k2 + 3) / 4 : 0, lay->kp);
  w.nsplit = pick_split(lay, w.nvt, nw);
  w.nchunk = (int)((w.nvt + SLAB_CHUNK - 1) / SLAB_CHUNK);
  size_t off = 0;
  auto take = [&](size_t elems) {
    double *ptr = base ? (double *)((char *)base + off) : nullptr;
    off += (elems * sizeof(double) + 255) / 256 * 256;
    return ptr;
  };
  w.mom_part = boot ? take((size_t)2 * w.nsplit * p * lay->k) : nullptr;
  w.opsum = boot ? take((size_t)4 * lay->nk * lay->k) : nullptr;
  w.sink = take(64);
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
                     width, w.nchunk, lay->kp, lay->k, lay->R, (int)lay_nquads(lay), lay_tpl(lay));
  return check_launch();
}

using ProjectKernel = void (*)(ProjectArgs);

int launch_kernel(ProjectKernel kern, int mode, const ProjectArgs &a, size_t lds, dim3 grid, int nw,
                  hipStream_t st) {
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void *)kern,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      g_last_hip = (int)e;
      return PLSR_ELAUNCH;
    }
  }
  TimedLaunch tl{};
  if (g_timing) {
    (void)hipEventCreate(&tl.a);
    (void)hipEventCreate(&tl.b);
    tl.kind = mode;
    (void)hipEventRecord(tl.a, st);
  }
  hipLaunchKernelGGL(kern, grid, dim3(64 * nw), lds, st, a);
  if (g_timing) {
    (void)hipEventRecord(tl.b, st);
    g_timed.push_back(tl);
  }
  return check_launch();
}

// Bootstrap instances: the period of the quad layout and the number of
// four-cell halves of the second matrix are compile-time in the hot kernel.
template <int P>
ProjectKernel boot_instance(int nh) {
  switch (nh) {
    case 0: return project_kernel<P, 1, 0>;
    case 1: return project_kernel<P, 1, 1>;
    case 2: return project_kernel<P, 1, 2>;
    case 3: return project_kernel<P, 1, 3>;
    default: return project_kernel<P, 1, 4>;
  }
}

template <int MODE>
int launch_project(const ProjectArgs &a, int period, int64_t nvt, int nsplit, hipStream_t st) {
  const int nw = lds_fed_waves(a.nk, period, MODE != 0, (a.k2 + 3) / 4, a.kp, MODE);
  size_t lds = project_lds_bytes(a.nk, period, MODE != 0, (a.k2 + 3) / 4, a.kp, nw);
#if PLSR_ABLATE & 256
  if (MODE == 0) lds += 40 * 1024;   // dev: force the permutation kernel down to two workgroups per CU
#endif
  dim3 grid((unsigned)nvt, (unsigned)nsplit);
  ProjectKernel kern = nullptr;
  if (MODE == 0) {
    kern = project_kernel<1, 0, 0>;
  } else {
    const int nh = (a.k2 + 3) / 4;
    switch (period) {
      case 1: kern = MODE == 1 ? boot_instance<1>(nh) : project_kernel<1, 2, -1>; break;
      case 2: kern = MODE == 1 ? boot_instance<2>(nh) : project_kernel<2, 2, -1>; break;
      case 3: kern = MODE == 1 ? boot_instance<3>(nh) : project_kernel<3, 2, -1>; break;
      case 4: kern = MODE == 1 ? boot_instance<4>(nh) : project_kernel<4, 2, -1>; break;
      case 5: kern = MODE == 1 ? boot_instance<5>(nh) : project_kernel<5, 2, -1>; break;
      case 6: kern = MODE == 1 ? boot_instance<6>(nh) : project_kernel<6, 2, -1>; break;
      default: return PLSR_EUNSUPPORTED;
    }
  }
  return launch_kernel(kern, MODE, a, lds, grid, nw, st);
}
template <bool DUMP, int NHT>
ProjectKernel boot_reg_instance(int nk) {
  switch (nk) {
    case 4: return project_boot_reg_kernel<4, DUMP, NHT>;
    case 5: return project_boot_reg_kernel<5, DUMP, NHT>;
    case 6: return project_boot_reg_kernel<6, DUMP, NHT>;
    case 7: return project_boot_reg_kernel<7, DUMP, NHT>;
    case 8: return project_boot_reg_kernel<8, DUMP, NHT>;
    case 9: return project_boot_reg_kernel<9, DUMP, NHT>;
    case 10: return project_boot_reg_kernel<10, DUMP, NHT>;
    case 11: return project_boot_reg_kernel<11, DUMP, NHT>;
    case 12: return project_boot_reg_kernel<12, DUMP, NHT>;
    case 13: return project_boot_reg_kernel<13, DUMP, NHT>;
    case 14: return project_boot_reg_kernel<14, DUMP, NHT>;
    case 15: return project_boot_reg_kernel<15, DUMP, NHT>;
    case 16: return project_boot_reg_kernel<16, DUMP, NHT>;
    default: return nullptr;
  }
}

// One-dimensional grid in XCD-aware order (reg_tile_and_run) for the kernels in `level`:
// 1 = K1r (default: its fetches drop from 705 to 89 MB per launch at config 2, same duration),
// 2 = K1br as well (fetches 665 -> 461 MB, but 1 % slower: with every run of a tile active at once the
// 2.9 MB of operator fragments compete with 1.1 GB of slab stores for the XCD's 4 MB of L2).
// PLSR_XCD_ORDER=0 / 1 / 2 overrides in developer builds (-DPLSR_DEV_KNOBS; measurement knob).
dim3 reg_grid(ProjectArgs &a, int64_t nvt, int nrun, int level) {
  static const int enabled = [] {
#ifdef PLSR_DEV_KNOBS                 // developer builds only: a stray variable must not change kernel selection
    if (const char *e = getenv("PLSR_XCD_ORDER")) {
      const int v = atoi(e);
      if (v >= 0 && v <= 2) return v;
    }
#endif
    return 1;
  }();
  const int64_t nwg = (nvt + 7) / 8 * 8 * nrun;
  if (enabled < level || nwg > 0x7fffffff) return dim3((unsigned)nvt, (unsigned)nrun);
  a.xcd_runs = nrun;
  a.nvt = nvt;
  return dim3((unsigned)nwg);
}

int launch_boot_reg(ProjectArgs a, int64_t nvt, hipStream_t st) {
  const int nh = (a.k2 + 3) / 4;
  ProjectKernel kern = a.vs_dump ? boot_reg_instance<true, -1>(a.nk)
                       : nh == 1 ? boot_reg_instance<false, 1>(a.nk)
                       : nh == 2 ? boot_reg_instance<false, 2>(a.nk)
                                 : boot_reg_instance<false, -1>(a.nk);
  if (!kern) return PLSR_EUNSUPPORTED;
  const size_t lds = ((size_t)((a.k2 + 3) / 4) * 4 * XM_LD + 16 * 64) * sizeof(double);
  TimedLaunch tl{};
  if (g_timing) {
    (void)hipEventCreate(&tl.a);
    (void)hipEventCreate(&tl.b);
    tl.kind = 1;
    (void)hipEventRecord(tl.a, st);
  }
  const dim3 grid = reg_grid(a, nvt, a.k * a.msplit, 2);
  hipLaunchKernelGGL(kern, grid, dim3(64), lds, st, a);
  if (g_timing) {
    (void)hipEventRecord(tl.b, st);
    g_timed.push_back(tl);
  }
  return check_launch();
}

// K1r: one wave per workgroup; the batch is split so that the grid has about
// twenty rounds of the 2048 resident waves (a wave's X fragments cost 4 nk loads
// per run, so runs should stay tens of tiles long)
int perm_reg_split(int ntiles, int64_t nvt) {
  const int nsplit = (int)std::min<int64_t>(ntiles, std::max<int64_t>(1, (20 * 2048 + nvt - 1) / nvt));
  return std::min(nsplit, std::max(1, ntiles / 8));
}

int launch_perm_reg(ProjectArgs a, int64_t nvt, hipStream_t st) {
  const int nsplit = perm_reg_split(a.ntiles, nvt);
  ProjectKernel kern = nullptr;
  switch (a.nk) {
    case 4: kern = project_perm_reg_kernel<4>; break;
    case 5: kern = project_perm_reg_kernel<5>; break;
    case 6: kern = project_perm_reg_kernel<6>; break;
    case 7: kern = project_perm_reg_kernel<7>; break;
    case 8: kern = project_perm_reg_kernel<8>; break;
    case 9: kern = project_perm_reg_kernel<9>; break;
    case 10: kern = project_perm_reg_kernel<10>; break;
    case 11: kern = project_perm_reg_kernel<11>; break;
    case 12: kern = project_perm_reg_kernel<12>; break;
    case 13: kern = project_perm_reg_kernel<13>; break;
    case 14: kern = project_perm_reg_kernel<14>; break;
    case 15: kern = project_perm_reg_kernel<15>; break;
    case 16: kern = project_perm_reg_kernel<16>; break;
    default: return PLSR_EUNSUPPORTED;
  }
  TimedLaunch tl{};
  if (g_timing) {
    (void)hipEventCreate(&tl.a);
    (void)hipEventCreate(&tl.b);
    tl.kind = 0;
    (void)hipEventRecord(tl.a, st);
  }
  const dim3 grid = reg_grid(a, nvt, nsplit, 1);
  hipLaunchKernelGGL(kern, grid, dim3(64), 0, st, a);
  if (g_timing) {
    (void)hipEventRecord(tl.b, st);
    g_timed.push_back(tl);
  }
  return check_launch();
}
}  // namespace

extern "C" size_t plsr_batch_workspace_bytes(const plsr_layout_t *lay, int64_t p, int32_t k2) {
  if (!lay || p <= 0 || k2 < 0) return 0;
  return carve(lay, p, k2, nullptr, true).bytes;
}

extern "C" int plsr_batch_plan(const plsr_layout_t *lay, int64_t p, int32_t k2, int32_t boot, int32_t out[4]) {
  if (!lay || !out || p <= 0 || k2 < 0) return PLSR_EINVAL;
  const Work w = carve(lay, p, k2, nullptr, boot != 0);
  const bool reg = lay->period == 0;
  out[0] = (int32_t)w.nvt;
  if (reg && !boot) {
    out[1] = perm_reg_split(lay->ntiles, w.nvt);          // K1r
  } else {
    out[1] = w.nsplit;
  }
  out[2] = reg ? lay_tpl(lay) : lay->ntiles;
  out[3] = reg ? 1 : 0;
  return PLSR_OK;
}

static int fill_common(ProjectArgs &a, const double *d_X, int64_t ldx, int64_t p,
                       const double *d_frag, const plsr_layout_t *lay) {
  if (!d_X || !d_frag || !lay || p <= 0 || ldx < p) return PLSR_EINVAL;
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
  Work w = carve(lay, p, 0, d_work, false);
  if (w.bytes > work_bytes) return PLSR_EWORKSPACE;
  a.norm_part = w.norm_part;
  a.sink = w.sink;
  hipStream_t st = (hipStream_t)stream;
  if (a.nk >= 4 && a.nk <= 16) {
    // n <= 64: the X fragments of a wave fit its registers (K1r)
    rc = launch_perm_reg(a, w.nvt, st);
  } else {
    // the permutation kernel keeps no per-latent-variable registers, so the
    // period-1 instance serves every k
    rc = launch_project<0>(a, 1, w.nvt, w.nsplit, st);
  }
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
  if (lay->period < 0 || lay->period > MAX_PERIOD) return PLSR_EUNSUPPORTED;
  Work w = carve(lay, p, k2, d_work, true);
  if (w.bytes > work_bytes) return PLSR_EWORKSPACE;
  a.norm_part = w.norm_part;
  a.T_part = w.T_part;
  a.sink = w.sink;
  a.ref = d_ref;
  a.Xm = k2 > 0 ? d_Xm : nullptr;
  a.ldxm = ldxm;
  a.k2 = k2;
  a.S1 = w.mom_part;
  a.S2 = w.mom_part + (size_t)w.nsplit * p * lay->k;
  a.vs_dump = d_vs_dump;
  hipStream_t st = (hipStream_t)stream;
  if (g_tail) {
    // the previous batch's reductions (tail stream) read the same workspace
    rc = chain(g_tail, st);
    if (rc) return rc;
  }
  const bool reg = lay->period == 0;
  if (reg) {
    // K1br: plain moment partials P1 [p][k], P2 [splits][p][k]; summed operator first
    a.S1 = w.mom_part;
    a.S2 = w.mom_part + (size_t)p * lay->k;
    a.tpl = lay_tpl(lay);
    a.msplit = w.nsplit;
    a.opsum = w.opsum;
    a.sink = w.sink;
    hipLaunchKernelGGL(opsum_kernel, dim3((unsigned)lay->nk, (unsigned)lay->k), dim3(256), 0, st, d_frag, w.opsum,
                       lay->nk, lay->k, a.tpl);
    rc = launch_boot_reg(a, w.nvt, st);
  } else {
    rc = d_vs_dump ? launch_project<2>(a, lay->period, w.nvt, w.nsplit, st)
                   : launch_project<1>(a, lay->period, w.nvt, w.nsplit, st);
  }
  if (rc) return rc;
  if (g_tail) {
    // merges and slab reductions follow the projection on the tail stream
    rc = chain(st, g_tail);
    if (rc) return rc;
    st = g_tail;
  }
  if (reg) {
    const int64_t cnt = p * lay->k;
    dim3 g((unsigned)((cnt + 255) / 256));
    hipLaunchKernelGGL(moment_shift_merge_kernel, g, dim3(256), 0, st, d_S1, d_S2, (const double *)a.S1,
                       (const double *)a.S2, d_ref, cnt, w.nsplit, (double)lay->R);
  } else {
    const int64_t cnt = p * lay->k;
    dim3 g((unsigned)((cnt + 255) / 256));
    hipLaunchKernelGGL(moment_merge_kernel, g, dim3(256), 0, st, d_S1, (const double *)a.S1, cnt,
                       w.nsplit);
    hipLaunchKernelGGL(moment_merge_kernel, g, dim3(256), 0, st, d_S2, (const double *)a.S2, cnt,
                       w.nsplit);
  }
  rc = reduce_slabs(w.norm_part, w, 1, lay, d_ssq, st);
  if (rc) return rc;
  if (k2 > 0) rc = reduce_slabs(w.T_part, w, k2, lay, d_T, st);
  return rc;
}

extern "C" int plsr_scale_cols(const double *d_in, int64_t rows, int32_t cols, const double *d_scale,
                               double *d_out, void *stream) {
  if (!d_in || !d_scale || !d_out || rows <= 0 || cols <= 0) return PLSR_EINVAL;
  const int64_t count = rows * cols;
  dim3 grid((unsigned)((count + 255) / 256));
  hipLaunchKernelGGL(scale_cols_kernel, grid, dim3(256), 0, (hipStream_t)stream, d_in, d_scale, count, cols, d_out);
  return check_launch();
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

extern "C" int plsr_set_tail_stream(void *stream) {
  g_tail = (hipStream_t)stream;
  return PLSR_OK;
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

// ---------------------------------------------------------------------------
// K2: Gram + eigen-decomposition
// ---------------------------------------------------------------------------
#include "plsr_gram.hip.h"

namespace {
struct GramPlan {
  int MC, B, MM, ks;
  int ngroups, nchunk, tiles_per_chunk;
  int64_t nvt;
  size_t lds, part_elems, bytes;
};

bool gram_plan(int32_t n, int32_t m, int32_t items, int64_t p, bool per_item_x, GramPlan &g) {
  if (n <= 0 || m <= 0 || items <= 0 || p <= 0) return false;
  const int nk = (n + 3) / 4;
  g.MC = (m + 15) / 16;
  if (g.MC > 6) return false;
  g.MM = g.MC * 16;
  const int cand1[] = {8, 4, 2, 1}, cand2[] = {2, 1}, cand3[] = {1};
  const int *cand = g.MC == 1 ? cand1 : (g.MC == 2 ? cand2 : cand3);
  const int ncand = g.MC == 1 ? 4 : (g.MC == 2 ? 2 : 1);
  g.B = 0;
  for (int i = 0; i < ncand; ++i) {
    if (cand[i] > 1 && (cand[i] > items || per_item_x)) continue;   // items with their own X do not share a tile
    // operator fragments of several items sharing a tile may take at most 96 KiB;
    // a single item may take up to 136 KiB (the rest stages rows of X in chunks)
    if ((size_t)cand[i] * g.MC * nk * 512 <= (size_t)(cand[i] > 1 ? 96 : 136) * 1024) {
      g.B = cand[i];
      break;
    }
  }
  if (g.B == 0) return false;
  {
    const size_t ops = (size_t)g.B * g.MC * nk * 512;
    const int fit = (int)((160 * 1024 - ops) / 2048);
    // a thread parks ks rows of the next chunk in registers (gram_pf: fewer for the six-tile
    // items with their own data, whose 21 Gram tiles leave less room)
    g.ks = std::min(std::min(nk, fit), gram_pf(g.MC, per_item_x));
    if (g.ks < 1) return false;
  }
  g.lds = gram_lds_bytes(nk, g.MC, g.B, g.ks);
  g.ngroups = (items + g.B - 1) / g.B;
  g.nvt = (p + TV - 1) / TV;
  int want = (int)std::max<int64_t>(1, (1024 + g.ngroups - 1) / g.ngroups);
  want = (int)std::min<int64_t>(want, g.nvt);
  g.tiles_per_chunk = (int)((g.nvt + want - 1) / want);
  g.nchunk = (int)((g.nvt + g.tiles_per_chunk - 1) / g.tiles_per_chunk);
  g.part_elems = (size_t)g.nchunk * items * g.MM * g.MM;
  g.bytes = (g.part_elems * sizeof(double) + 255) / 256 * 256;
  return true;
}

template <int MC, int B, bool FUSED = false, int SL = 0, int SH = MC>
int launch_gram(const GramArgs &a, const GramPlan &g, hipStream_t st) {
  auto kern = gram_kernel<MC, B, FUSED, SL, SH>;
  if (g.lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void *)kern,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds);
    if (e != hipSuccess) {
      g_last_hip = (int)e;
      return PLSR_ELAUNCH;
    }
  }
  hipLaunchKernelGGL(kern, dim3(g.ngroups, g.nchunk), dim3(256), g.lds, st, a);
  return check_launch();
}
}  // namespace

extern "C" int64_t plsr_rows_frag_elems(int32_t n, int32_t m, int32_t items) {
  if (n <= 0 || m <= 0 || items <= 0) return 0;
  return (int64_t)items * ((m + 15) / 16) * ((n + 3) / 4) * 64;
}

extern "C" int plsr_ops_pack_rows(const double *d_rows, int32_t items, int32_t m, int32_t n,
                                  double *d_frag, void *stream) {
  if (!d_rows || !d_frag || items <= 0 || m <= 0 || n <= 0) return PLSR_EINVAL;
  const int MC = (m + 15) / 16, nk = (n + 3) / 4;
  const int64_t total = (int64_t)items * MC * nk * 64;
  hipLaunchKernelGGL(ops_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, d_rows, d_frag, items, m, n, nk, MC);
  return check_launch();
}

extern "C" size_t plsr_gram_workspace_bytes(int32_t n, int32_t m, int32_t items, int64_t p,
                                            int64_t x_item_stride) {
  GramPlan g;
  return gram_plan(n, m, items, p, x_item_stride != 0, g) ? g.bytes : 0;
}

extern "C" int plsr_gram_batch(const double *d_X, int64_t x_item_stride, int64_t ldx, int64_t p,
                               int32_t n, const double *d_frag, int32_t items, int32_t m,
                               double *d_G, void *d_work, size_t work_bytes, void *stream) {
  if (!d_X || !d_frag || !d_G || !d_work || ldx < p || x_item_stride < 0) return PLSR_EINVAL;
  GramPlan g;
  if (!gram_plan(n, m, items, p, x_item_stride != 0, g)) return PLSR_EUNSUPPORTED;
  if (g.bytes > work_bytes) return PLSR_EWORKSPACE;
  GramArgs a;
  a.X = d_X;
  a.x_item_stride = x_item_stride;
  a.ldx = ldx;
  a.p = p;
  a.n = n;
  a.nk = (n + 3) / 4;
  a.frag = d_frag;
  a.items = items;
  a.tiles_per_chunk = g.tiles_per_chunk;
  a.ks = g.ks;
  a.G_part = (double *)d_work;
  a.src = nullptr;
  a.rowcell = nullptr;
  a.ncell = 0;
  a.sc = a.sh = nullptr;
  a.act = nullptr;
  a.rowtab = nullptr;
  hipStream_t st = (hipStream_t)stream;
  int rc = PLSR_EUNSUPPORTED;
#define PLSR_G(MCv, Bv) \
  if (g.MC == MCv && g.B == Bv) rc = launch_gram<MCv, Bv>(a, g, st);
  PLSR_G(1, 8) PLSR_G(1, 4) PLSR_G(1, 2) PLSR_G(1, 1) PLSR_G(2, 2) PLSR_G(2, 1)
  PLSR_G(3, 1) PLSR_G(4, 1) PLSR_G(5, 1) PLSR_G(6, 1)
#undef PLSR_G
  if (rc) return rc;
  // sum the voxel chunks (fixed order)
  const int64_t E = (int64_t)items * g.MM * g.MM;
  hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)((E + 255) / 256), 1), dim3(256), 0, st,
                     (const double *)d_work, d_G, E, g.nchunk, g.nchunk);
  return check_launch();
}

extern "C" int plsr_eigh_batch(const double *d_G, int64_t item_stride, int32_t ld, int32_t off,
                               int32_t k, int32_t count, double *d_evals, double *d_evecs,
                               const double *d_init, int32_t relative, void *stream) {
  if (!d_G || !d_evals || !d_evecs || count <= 0 || k <= 0 || ld < off + k) return PLSR_EINVAL;
  if (k > EIG_MAX) return PLSR_EUNSUPPORTED;
  if (d_init == d_evecs) return PLSR_EINVAL;          // the output is written in sorted column order
  hipLaunchKernelGGL(eigh_kernel, dim3((unsigned)count), dim3(64), eigh_lds_bytes(k), (hipStream_t)stream, d_G,
                     item_stride, ld, off, k, count, d_evals, d_evecs, 30, d_init, (int)(relative != 0));
  return check_launch();
}

extern "C" int plsr_rotate_rows(const double *d_U, const double *d_rows_in, double *d_rows_out, int32_t items,
                                int32_t m, int32_t n, int32_t off, int32_t k, void *stream) {
  if (!d_U || !d_rows_in || !d_rows_out || items <= 0 || m <= 0 || n <= 0 || off < 0 || k <= 0 || off + k > m ||
      d_rows_in == d_rows_out)
    return PLSR_EINVAL;
  hipLaunchKernelGGL(rotate_rows_kernel, dim3((unsigned)(((int64_t)m * n + 255) / 256), (unsigned)items), dim3(256), 0,
                     (hipStream_t)stream, d_U, d_rows_in, d_rows_out, m, n, off, k);
  return check_launch();
}

extern "C" int plsr_svd_finish(const double *d_lam, const double *d_cur, int32_t k, int32_t n, double abs_tol,
                               double rel_tol, double *d_s, double *d_rows_out, void *stream) {
  if (!d_lam || !d_cur || !d_s || !d_rows_out || k <= 0 || n <= 0) return PLSR_EINVAL;
  hipLaunchKernelGGL(svd_finish_kernel, dim3((unsigned)((k * n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     d_lam, d_cur, k, n, abs_tol, rel_tol, d_s, d_rows_out);
  return check_launch();
}

// ---------------------------------------------------------------------------
// K3: gather + per-cell z-score
// ---------------------------------------------------------------------------
#include "plsr_corr.hip.h"

extern "C" int plsr_gather_zscore(const double *d_X, int64_t ldx, int64_t p, const int32_t *d_src,
                                  int32_t items, int32_t nout, const int32_t *d_cell_lo,
                                  const int32_t *d_cell_z, int32_t ncell, double *d_out,
                                  int64_t ldo, void *stream) {
  if (!d_X || !d_src || !d_cell_lo || !d_cell_z || !d_out || items <= 0 || nout <= 0 ||
      ncell <= 0 || p <= 0 || ldx < p || ldo < p)
    return PLSR_EINVAL;
  GatherArgs a;
  a.X = d_X;
  a.ldx = ldx;
  a.p = p;
  a.src = d_src;
  a.cell_lo = d_cell_lo;
  a.cell_z = d_cell_z;
  a.nout = nout;
  a.ncell = ncell;
  a.out = d_out;
  a.ldo = ldo;
  hipLaunchKernelGGL(gather_zscore_kernel, dim3((unsigned)((p + 255) / 256), (unsigned)items),
                     dim3(256), 0, (hipStream_t)stream, a);
  return check_launch();
}

extern "C" int plsr_apply_rows(const double *d_X, int64_t ldx, int64_t p, int32_t n, const double *d_rows,
                               int32_t m, double *d_out, int64_t ldo, void *stream) {
  if (!d_X || !d_rows || !d_out || n <= 0 || m <= 0 || p <= 0 || ldx < p || ldo < p) return PLSR_EINVAL;
  if ((size_t)n * 16 * sizeof(double) > 64 * 1024) return PLSR_EUNSUPPORTED;       // n <= 512
  hipLaunchKernelGGL(rows_apply_kernel, dim3((unsigned)((p + 255) / 256), (unsigned)((m + 15) / 16)), dim3(256),
                     (size_t)n * 16 * sizeof(double), (hipStream_t)stream, d_X, ldx, p, n, d_rows, m, d_out, ldo);
  return check_launch();
}

// K4 / K5 entry points (same translation unit: the shared kernels in
// plsr_project.hip.h are defined once)
#include "plsr_item_abi.hip.h"
#include "plsr_split_abi.hip.h"
