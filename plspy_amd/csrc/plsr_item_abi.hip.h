// extern "C" entry points for K4f (fused item projection), the fused Gram and
// K5 (latent scores): bootstrap and split-half of behaviour / multiblock PLS.
// See include/plsr.h.
#include "../../include/plsr.h"
#include "plsr_item.hip.h"
#include "plsr_fused.hip.h"
#include "plsr_agg.hip.h"
#include "plsr_beh.hip.h"
#include "plsr_io.hip.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace plsr;

namespace {
int launch_ok() { return hipGetLastError() == hipSuccess ? PLSR_OK : PLSR_ELAUNCH; }

// items per workgroup of the latent kernel.  Items of a group share every staged X tile, but
// the accumulators (IG * MC * NI x 8 VGPRs) decide how many waves a SIMD holds: from three
// tiles of latent variables on (k > 32) a single item per workgroup -- 114 VGPRs, two
// workgroups per CU, four waves per SIMD -- beats three items at two waves per SIMD by 9 %
// (config 3: 133 -> 121 ms per 2000 items) although X is then staged once per item; below,
// where the X tile dominates the traffic, up to three items share it (more spilled).
constexpr int latent_group_cap(int mc, int ni) {
  if (mc >= 3) return 1;
  const int c = 10 / (mc * ni);
  return c < 1 ? 1 : (c > 3 ? 3 : c);
}

struct LatentPlan {
  int MC, NI, IG, WV, ngroups, nchunk, tiles_per_chunk;
  size_t lds, z_elems, n_elems, bytes;
};

// data rows whose 32-bit byte offsets from a block's first row stay below 4 GiB (a multiple of 16, or all n)
inline int latent_rows_per_block(int32_t n, int64_t ldx) {
  const int64_t fit = ((((int64_t)1 << 32) - 1024) / (ldx * 8));
  return fit >= n ? n : (int)(fit / 16 * 16);
}

bool latent_plan(int32_t n, int32_t k, int32_t items, int64_t p, LatentPlan &pl) {
  if (n <= 0 || k <= 0 || items <= 0 || p <= 0) return false;
  pl.MC = (k + 15) / 16;
  // eight waves (one tile of data rows each) when there are that many tiles: small
  // accumulators, four waves per SIMD at two workgroups per CU
  pl.WV = (n + 15) / 16 > 4 ? 8 : 4;
#ifdef PLSR_DEV_KNOBS                                               // developer builds only (-DPLSR_DEV_KNOBS, through PLSR_LIB)
  if (const char *e = getenv("PLSR_LATENT_WV")) {
    const int wv = atoi(e);
    if (wv == 4 || wv == 8) pl.WV = wv;
  }
#endif
  pl.NI = ((n + 15) / 16 + pl.WV - 1) / pl.WV;
  if (pl.MC > 4 || pl.NI > 2) return false;
  // items per workgroup: as many as the accumulators allow (IG * MC * NI tiles of 8 VGPRs)
  pl.IG = latent_group_cap(pl.MC, pl.NI);
  pl.IG = std::min(pl.IG, (int)items);
  pl.ngroups = (items + pl.IG - 1) / pl.IG;
  pl.lds = latent_lds_bytes(pl.MC, pl.NI, pl.IG, pl.WV);
  const int64_t nvt = (p + LV_T - 1) / LV_T;
  // voxel chunks: about 1024 workgroups, and a count whose last round of the chip's resident
  // workgroups (two per CU with eight waves, LDS permitting) is as full as the others -- 125 groups
  // x 9 chunks were 2.2 rounds of 512 and ran as three
  {
    const int64_t slots = 512;
    const int lo = (int)std::max<int64_t>(1, (768 + pl.ngroups - 1) / pl.ngroups);
    const int hi = (int)std::max<int64_t>(lo, (2048 + pl.ngroups - 1) / pl.ngroups);
    int want = lo;
    double best = -1.0;
    for (int c = lo; c <= hi; ++c) {
      const double rounds = (double)pl.ngroups * c / (double)slots;
      const double eff = rounds / std::ceil(rounds);
      if (eff > best + 1e-3) {
        best = eff;
        want = c;
      }
    }
    want = (int)std::min<int64_t>(want, nvt);
    pl.tiles_per_chunk = (int)((nvt + want - 1) / want);
    pl.nchunk = (int)((nvt + pl.tiles_per_chunk - 1) / pl.tiles_per_chunk);
  }
  pl.z_elems = (size_t)pl.nchunk * items * k * n;
  pl.n_elems = (size_t)pl.nchunk * items * k;
  pl.bytes = ((pl.z_elems + pl.n_elems) * sizeof(double) + 511) / 256 * 256;
  // The kernel addresses the rows a thread stages by 32-bit byte offsets from a group's first item (VS^T) and
  // from X (assuming the rows are p apart -- the call checks its actual strides).  VS^T of a group must fit;
  // an X of 4 GiB or more (n = 240 at p >= 2.24 M) is walked in blocks of rows, each below 4 GiB.
  if ((int64_t)pl.IG * k * p * 8 >= ((int64_t)1 << 32)) return false;
  if (latent_rows_per_block(n, p) < 16 && latent_rows_per_block(n, p) < n) return false;
  return true;
}

// sums the voxel chunks of a block of data rows into its columns of Zt:
// out[jk][row_lo + i] = sum_c part[c][jk][i], jk = (item, latent variable), i < nb
__global__ __launch_bounds__(256) void latent_sum_rows_kernel(const double *part, double *out, int64_t E, int nb, int n,
                                                              int row_lo, int nchunk) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= E * nb) return;
  const int64_t jk = e / nb;
  const int i = (int)(e - jk * nb);
  double acc = 0.0;
  for (int c = 0; c < nchunk; ++c) acc += part[(int64_t)c * E * nb + e];
  out[jk * n + row_lo + i] = acc;
}

template <int MC, int NI, int IG, int WV>
int run_latent_wv(const LatentArgs &a, const LatentPlan &pl, hipStream_t st) {
  auto kern = latent_kernel<MC, NI, IG, WV>;
  if (pl.lds > 64 * 1024 &&
      hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds) != hipSuccess)
    return PLSR_ELAUNCH;
  hipLaunchKernelGGL(kern, dim3((unsigned)pl.ngroups, (unsigned)pl.nchunk), dim3(WV * 64), pl.lds, st, a);
  return launch_ok();
}

template <int MC, int NI, int IG>
int run_latent_ig(const LatentArgs &a, const LatentPlan &pl, hipStream_t st) {
  return pl.WV == 8 ? run_latent_wv<MC, NI, IG, 8>(a, pl, st) : run_latent_wv<MC, NI, IG, 4>(a, pl, st);
}

template <int MC, int NI>
int run_latent(const LatentArgs &a, const LatentPlan &pl, hipStream_t st) {
  constexpr int cap = latent_group_cap(MC, NI);
  if constexpr (cap >= 3) {
    if (pl.IG >= 3) return run_latent_ig<MC, NI, 3>(a, pl, st);
  }
  if constexpr (cap >= 2) {
    if (pl.IG >= 2) return run_latent_ig<MC, NI, 2>(a, pl, st);
  }
  return run_latent_ig<MC, NI, 1>(a, pl, st);
}
}  // namespace

extern "C" size_t plsr_latent_workspace_bytes(int32_t n, int32_t k, int32_t items, int64_t p) {
  LatentPlan pl;
  return latent_plan(n, k, items, p, pl) ? pl.bytes : 0;
}

extern "C" int plsr_latent(const double *d_X, int64_t ldx, int64_t p, int32_t n, const double *d_vst,
                           int64_t ldv, int32_t items, int32_t k, double *d_Zt, double *d_nsq,
                           void *d_work, size_t work_bytes, void *stream) {
  if (!d_X || !d_vst || !d_Zt || !d_work || ldx < p || ldv < p) return PLSR_EINVAL;
  LatentPlan pl;
  if (!latent_plan(n, k, items, p, pl)) return PLSR_EUNSUPPORTED;
  if (pl.bytes > work_bytes) return PLSR_EWORKSPACE;
  if ((int64_t)pl.IG * k * ldv * 8 >= ((int64_t)1 << 32)) return PLSR_EUNSUPPORTED;   // (a stride wider than p)
  const int nb_max = latent_rows_per_block(n, ldx);
  if (nb_max < n && nb_max < 16) return PLSR_EUNSUPPORTED;
  LatentArgs a;
  a.ldx = ldx;
  a.p = p;
  a.k = k;
  a.items = items;
  a.vst = d_vst;
  a.ldv = ldv;
  a.tiles_per_chunk = pl.tiles_per_chunk;
  a.Zt_part = (double *)d_work;
  hipStream_t st = (hipStream_t)stream;
  // blocks of data rows (one block unless X is 4 GiB or more): Zt's columns row_lo .. row_lo + nb
  for (int row_lo = 0; row_lo < n; row_lo += nb_max) {
    const int nb = std::min(nb_max, n - row_lo);
    a.X = d_X + (int64_t)row_lo * ldx;
    a.n = nb;
    a.nsq_part = (d_nsq && row_lo == 0) ? a.Zt_part + pl.z_elems : nullptr;     // the column norms of VS: once
    int rc = PLSR_EUNSUPPORTED;
#define PLSR_L(M, N) \
  if (pl.MC == M && pl.NI == N) rc = run_latent<M, N>(a, pl, st);
    PLSR_L(1, 1) PLSR_L(1, 2) PLSR_L(2, 1) PLSR_L(2, 2) PLSR_L(3, 1) PLSR_L(3, 2) PLSR_L(4, 1) PLSR_L(4, 2)
#undef PLSR_L
    if (rc) return rc;
    const int64_t E = (int64_t)items * k;
    hipLaunchKernelGGL(latent_sum_rows_kernel, dim3((unsigned)((E * nb + 255) / 256)), dim3(256), 0, st,
                       (const double *)a.Zt_part, d_Zt, E, nb, n, row_lo, pl.nchunk);
    if (a.nsq_part)
      hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)((E + 255) / 256), 1), dim3(256), 0, st,
                         (const double *)a.nsq_part, d_nsq, E, pl.nchunk, pl.nchunk);
  }
  return launch_ok();
}

extern "C" size_t plsr_latent_xt_bytes(int32_t n, int64_t p) {
  if (n <= 0 || n > XT_LD || p <= 0) return 0;
  return (size_t)((p + LV_T - 1) / LV_T * LV_T) * XT_LD * sizeof(double);
}

extern "C" int plsr_latent_xt_prepare(const double *d_X, int64_t ldx, int64_t p, int32_t n, double *d_XT, void *stream) {
  if (!d_X || !d_XT || n <= 0 || n > XT_LD || p <= 0 || ldx < p) return PLSR_EINVAL;
  const int64_t p_pad = (p + LV_T - 1) / LV_T * LV_T;
  hipLaunchKernelGGL(xt_prepare_kernel, dim3((unsigned)(p_pad / 32), XT_LD / 32), dim3(256), 0, (hipStream_t)stream, d_X,
                     ldx, p, n, d_XT, p_pad);
  return launch_ok();
}

namespace {
// chunks of voxel tiles for a launch of `items` workgroups that fit `slots` at a time: the last round of resident
// workgroups as full as the others
void latent_xt_chunks(int32_t items, int64_t p, int slots, int32_t k, int32_t ncol, LatentPlan &pl) {
  const int64_t nvt = (p + LV_T - 1) / LV_T;
  const int lo = (int)std::max<int64_t>(1, (768 + items - 1) / items);
  const int hi = (int)std::max<int64_t>(lo, (2048 + items - 1) / items);
  int want = lo;
  double best = -1.0;
  for (int c = lo; c <= hi; ++c) {
    const double rounds = (double)items * c / (double)slots;
    const double eff = rounds / std::ceil(rounds);
    if (eff > best + 1e-3) {
      best = eff;
      want = c;
    }
  }
  want = (int)std::min<int64_t>(want, nvt);
  pl.tiles_per_chunk = (int)((nvt + want - 1) / want);
  pl.nchunk = (int)((nvt + pl.tiles_per_chunk - 1) / pl.tiles_per_chunk);
  pl.z_elems = (size_t)pl.nchunk * items * k * ncol;
  pl.n_elems = (size_t)pl.nchunk * items * k;
  pl.bytes = ((pl.z_elems + pl.n_elems) * sizeof(double) + 511) / 256 * 256;
}

int latent_xt_launch(const LatentXtArgs &a, int MC, int WV, int nchunk, hipStream_t st) {
  const size_t lds = std::max((size_t)2 * MC * 16 * XT_VLD, (size_t)8 * MC * 16) * sizeof(double);
  int rc = PLSR_EUNSUPPORTED;
#define PLSR_LX(M, W)                                                                                              \
  if (MC == M && WV <= W && rc == PLSR_EUNSUPPORTED) {                                                             \
    hipLaunchKernelGGL((latent_xt_kernel<M, W>), dim3((unsigned)a.items, (unsigned)nchunk), dim3(W * 64), lds, st, \
                       a);                                                                                         \
    rc = launch_ok();                                                                                              \
  }
  PLSR_LX(1, 4) PLSR_LX(1, 8) PLSR_LX(2, 4) PLSR_LX(2, 8) PLSR_LX(3, 4) PLSR_LX(3, 8) PLSR_LX(4, 4) PLSR_LX(4, 8)
#undef PLSR_LX
  return rc;
}

// K5i: columns of the partial outputs, voxel ranges per item, workspace
inline int index_cols(int32_t max_rows) { return (max_rows + 15) / 16 * 16; }
inline size_t index_meta_bytes(int32_t items, int32_t m) {
  return (((size_t)items * (XT_LD + 1 + m)) * sizeof(int32_t) + 255) / 256 * 256;
}
// about four rounds of the chip's 2048 wave slots (two waves per SIMD), a multiple of 8, at least 16 tiles each
inline void index_splits(int32_t items, int64_t p, int &nsplit, int &tiles_per_split) {
  const int64_t nvt = (p + LV_T - 1) / LV_T;
  int64_t s = (8192 + (int64_t)items * 8 - 1) / ((int64_t)items * 8) * 8;
  s = std::min<int64_t>(s, std::max<int64_t>(8, nvt / 16 / 8 * 8));
  s = std::max<int64_t>(8, std::min<int64_t>(s, 512));
  tiles_per_split = (int)((nvt + s - 1) / s);
  nsplit = (int)s;
}
inline size_t index_part_bytes(int32_t items, int32_t k, int ncol, int nsplit) {
  return (((size_t)nsplit * items * k * (ncol + 1)) * sizeof(double) + 511) / 256 * 256;
}
}  // namespace

extern "C" int plsr_latent_xt(const double *d_XT, int64_t p, int32_t n, const double *d_vst, int64_t ldv,
                              int32_t items, int32_t k, double *d_Zt, double *d_nsq, void *d_work, size_t work_bytes,
                              void *stream) {
  if (!d_XT || !d_vst || !d_Zt || !d_work || ldv < p) return PLSR_EINVAL;
  if (n <= 0 || n > XT_LD || k <= 0 || k > 64) return PLSR_EUNSUPPORTED;
  LatentPlan pl;
  if (!latent_plan(n, k, items, p, pl)) return PLSR_EUNSUPPORTED;
  latent_xt_chunks(items, p, 512, k, n, pl);        // one item per workgroup here: chunks for `items` groups
  if (pl.bytes > work_bytes) return PLSR_EWORKSPACE;
  if ((int64_t)k * ldv * 8 >= ((int64_t)1 << 32)) return PLSR_EUNSUPPORTED;
  LatentXtArgs a;
  a.XT = d_XT;
  a.p = p;
  a.n = n;
  a.k = k;
  a.items = items;
  a.vst = d_vst;
  a.ldv = ldv;
  a.tiles_per_chunk = pl.tiles_per_chunk;
  a.Zt_part = (double *)d_work;
  a.nsq_part = d_nsq ? a.Zt_part + pl.z_elems : nullptr;
  hipStream_t st = (hipStream_t)stream;
  const int rc = latent_xt_launch(a, (k + 15) / 16, (n + 15) / 16, pl.nchunk, st);
  if (rc) return rc;
  const int64_t EZ = (int64_t)items * k * n, EN = (int64_t)items * k;
  hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)((EZ + 255) / 256), 1), dim3(256), 0, st,
                     (const double *)a.Zt_part, d_Zt, EZ, pl.nchunk, pl.nchunk);
  if (d_nsq)
    hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)((EN + 255) / 256), 1), dim3(256), 0, st,
                       (const double *)a.nsq_part, d_nsq, EN, pl.nchunk, pl.nchunk);
  return launch_ok();
}

extern "C" size_t plsr_latent_xt_workspace_bytes(int32_t n, int32_t k, int32_t items, int64_t p) {
  if (n <= 0 || n > XT_LD || k <= 0 || k > 64 || items <= 0 || p <= 0) return 0;
  const int64_t nvt = (p + LV_T - 1) / LV_T;
  const int64_t hi = std::min<int64_t>(nvt, std::max<int64_t>(1, (2048 + items - 1) / items));
  return (size_t)(((size_t)hi * items * k * (n + 1)) * sizeof(double) + 511) / 256 * 256;
}

// K5i: L_b = (X[idx_b] VS_b^T)^T from the rows of X the sample holds, each once
extern "C" size_t plsr_latent_xb_bytes(int32_t n, int64_t p) {
  if (n <= 0 || n > XT_LD || p <= 0) return 0;
  return (size_t)((p + LV_T - 1) / LV_T * LV_T) * n * sizeof(double);
}

extern "C" int plsr_latent_xb_prepare(const double *d_X, int64_t ldx, int64_t p, int32_t n, double *d_XB, void *stream) {
  if (!d_X || !d_XB || n <= 0 || n > XT_LD || p <= 0 || ldx < p) return PLSR_EINVAL;
  const int64_t E = (p + LV_T - 1) / LV_T * n * 16;
  hipLaunchKernelGGL(xb_prepare_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_X, ldx, p,
                     n, d_XB);
  return launch_ok();
}

extern "C" size_t plsr_latent_index_workspace_bytes(int32_t n, int32_t k, int32_t items, int64_t p, int32_t m,
                                                    int32_t max_rows, int32_t t_rows) {
  if (n <= 0 || n > XT_LD || k <= 0 || k > 64 || items <= 0 || p <= 0 || m <= 0 || max_rows <= 0 || max_rows > n)
    return 0;                                        // (the row lists hold up to XT_LD = 128 rows of X)
  if (t_rows < 0 || t_rows > 16) return 0;
  if ((int64_t)items * k * (m + t_rows) >= ((int64_t)1 << 31)) return 0;
  if ((int64_t)k * ((p + LV_T - 1) / LV_T * LV_T) * 8 >= ((int64_t)1 << 32)) return 0;   // 32-bit offsets into an item's VS^T
  int nsplit, tps;
  index_splits(items, p, nsplit, tps);
  return index_meta_bytes(items, m) + index_part_bytes(items, k, index_cols(max_rows) + (t_rows ? 16 : 0), nsplit);
}

extern "C" int plsr_latent_index(const double *d_XB, int64_t p, int32_t n, const double *d_vst, int64_t ldv,
                                 int32_t vst_tiled, int32_t items, int32_t k, const int32_t *d_idx, int32_t m,
                                 int32_t max_rows, const double *d_T, int64_t t_ld, int32_t t_item_rows,
                                 const int32_t *d_t_row, int32_t t_rows, double *d_L, double *d_nsq, void *d_work,
                                 size_t work_bytes, void *stream) {
  if (!d_XB || !d_vst || !d_L || !d_work || !d_idx || ldv < p) return PLSR_EINVAL;
  if (vst_tiled && ldv % LV_T != 0) return PLSR_EINVAL;
  if (!d_T) t_rows = 0;
  if (t_rows && (!d_t_row || t_ld < p || t_item_rows <= 0)) return PLSR_EINVAL;
  const size_t need = plsr_latent_index_workspace_bytes(n, k, items, p, m, max_rows, t_rows);
  if (!need) return PLSR_EUNSUPPORTED;
  if (need > work_bytes) return PLSR_EWORKSPACE;
  if ((int64_t)k * ldv * 8 >= ((int64_t)1 << 32) || (int64_t)n * LV_T * 8 >= ((int64_t)1 << 32)) return PLSR_EUNSUPPORTED;
  if (t_rows && (int64_t)t_item_rows * t_ld * 8 >= ((int64_t)1 << 32)) return PLSR_EUNSUPPORTED;
  LatentWaveArgs a;
  index_splits(items, p, a.nsplit, a.tiles_per_split);
  if ((int64_t)items * a.nsplit >= ((int64_t)1 << 31)) return PLSR_EUNSUPPORTED;
  const size_t meta = index_meta_bytes(items, m);
  int32_t *rows = (int32_t *)d_work;
  int32_t *nu = rows + (size_t)items * XT_LD;
  int32_t *cols = nu + items;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(latent_index_meta_kernel, dim3((unsigned)items), dim3(128), 0, st, d_idx, (int)m, (int)n,
                     (int)max_rows, rows, nu, cols);
  a.XB = d_XB;
  a.p = p;
  a.n = n;
  a.k = k;
  a.items = items;
  a.vst = d_vst;
  a.ldv = ldv;
  a.vst_tiled = vst_tiled != 0;
  a.nr = index_cols(max_rows) + (t_rows ? 16 : 0);
  a.Zt_part = (double *)((char *)d_work + meta);
  a.nsq_part = d_nsq ? a.Zt_part + (size_t)a.nsplit * items * k * a.nr : nullptr;
  a.rows = rows;
  a.nu = nu;
  a.T = t_rows ? d_T : nullptr;
  a.t_ld = t_ld;
  a.t_item_rows = t_item_rows;
  a.t_rows = t_rows;
  a.t_row = d_t_row;
  const int MC = (k + 15) / 16;
  const int tiles = index_cols(max_rows) / 16 + (t_rows ? 1 : 0);
  const unsigned grid = (unsigned)((int64_t)items * a.nsplit);
  int rc = PLSR_EUNSUPPORTED;
#define PLSR_LW(M)                                                                            \
  if (MC == M) {                                                                              \
    a.tile_lo = 0;                                                                            \
    hipLaunchKernelGGL((latent_wave_kernel<M>), dim3(grid), dim3(64), 0, st, a);              \
    if (tiles > latent_wave_cap(M)) { /* the tiles past a wave's capacity */                  \
      a.tile_lo = latent_wave_cap(M);                                                         \
      a.nsq_part = nullptr;                                                                   \
      hipLaunchKernelGGL((latent_wave_kernel<M>), dim3(grid), dim3(64), 0, st, a);            \
    }                                                                                         \
    rc = launch_ok();                                                                         \
  }
  double *const nsq_part = a.nsq_part;
  PLSR_LW(1) PLSR_LW(2) PLSR_LW(3) PLSR_LW(4)
  a.nsq_part = nsq_part;
#undef PLSR_LW
  if (rc) return rc;
  const int64_t EL = (int64_t)items * k * (m + t_rows), EN = (int64_t)items * k;
  hipLaunchKernelGGL(latent_index_sum_kernel, dim3((unsigned)((EL + 255) / 256)), dim3(256), 0, st,
                     (const double *)a.Zt_part, (int64_t)items, (int)k, (int)a.nr, (int)a.nsplit,
                     (const int32_t *)cols, (const int32_t *)nu, (int)m, (int)t_rows, d_L);
  if (d_nsq)
    hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)((EN + 255) / 256), 1), dim3(256), 0, st,
                       (const double *)a.nsq_part, d_nsq, EN, a.nsplit, a.nsplit);
  return launch_ok();
}

// ---------------------------------------------------------------------------
// K4f: fused gather / z-score / projection (plsr_fused.hip.h)
// ---------------------------------------------------------------------------
namespace {
struct FusedPlan {
  FusedCells cells;    // the cells of the statistics (caller's cells)
  FusedCells kcells;   // the cells item_fused2_kernel walks: pieces of at most FZ_CELL_STEPS k-steps
  int MC, NT, TVX, VB, waves, nsplit, nchunk, flat, nslabm;
  int64_t nvt, nslab;
  size_t lds;
  // workspace carve (byte offsets)
  size_t o_frag, o_rowoff, o_sc, o_sh, o_mom, o_sq, o_sq2, bytes;
};

bool fused_plan(int32_t n, int32_t nz, int32_t k, const int32_t *cell_lo, const int32_t *cell_z,
                int32_t ncell, int32_t items, int64_t p, bool moments, bool rowsq, FusedPlan &pl) {
  if (n <= 0 || nz <= 0 || k <= 0 || items <= 0 || p <= 0 || !cell_lo || ncell <= 0 || ncell > FZ_MAXCELL)
    return false;
  if (cell_lo[0] != 0 || cell_lo[ncell] != nz) return false;
  pl.MC = (k + 15) / 16;
  if (pl.MC > 8) return false;
  // One wave = one 16-row tile of latent variables x NT 16-voxel tiles; a
  // fragment load feeds NT MFMAs, so NT = 4 keeps the L2 traffic at the level of
  // the projection kernel.  (Measured at config 3, k = 48: 64-voxel workgroups
  // of three NT = 4 waves beat 32-voxel ones with NT = 2 at five per CU, whose
  // vector L1 stalls on twice the fragment traffic.)
  pl.TVX = 64;
  pl.NT = pl.MC >= 3 ? 4 : pl.MC;
  pl.VB = pl.TVX / (16 * pl.NT);
  pl.waves = pl.MC * pl.VB;
  pl.flat = pl.MC == 3 ? 1 : 0;        // four waves share three tiles' worth of (tile, item) tasks
  if (pl.flat) pl.waves = 4;
  const size_t lds_x = (size_t)n * pl.TVX * sizeof(double);
  if ((size_t)n * TV * sizeof(double) > 160 * 1024) return false;      // the statistics kernel's tile
  pl.cells.ncell = ncell;
  int steps = 0;
  for (int c = 0; c < ncell; ++c) {
    if (cell_lo[c + 1] <= cell_lo[c]) return false;
    pl.cells.row_lo[c] = cell_lo[c];
    pl.cells.step_lo[c] = steps;
    pl.cells.z[c] = cell_z ? cell_z[c] : 1;
    pl.cells.stat[c] = c;
    steps += (cell_lo[c + 1] - cell_lo[c] + 3) / 4;
  }
  pl.cells.row_lo[ncell] = cell_lo[ncell];
  pl.cells.step_lo[ncell] = steps;
  pl.cells.nkp = steps;
  // kernel cells: every cell cut into pieces of FZ_CELL_STEPS k-steps (the cuts fall on whole
  // k-steps, so the padded step count does not change)
  pl.kcells = pl.cells;
  {
    int kc = 0, ks = 0;
    for (int c = 0; c < ncell; ++c) {
      for (int lo = cell_lo[c]; lo < cell_lo[c + 1]; lo += 4 * FZ_CELL_STEPS) {
        if (kc >= FZ_MAXPIECE) return false;
        const int hi = std::min(cell_lo[c + 1], lo + 4 * FZ_CELL_STEPS);
        pl.kcells.row_lo[kc] = lo;
        pl.kcells.step_lo[kc] = ks;
        pl.kcells.z[kc] = pl.cells.z[c];
        pl.kcells.stat[kc] = c;
        ks += (hi - lo + 3) / 4;
        ++kc;
      }
    }
    pl.kcells.ncell = kc;
    pl.kcells.row_lo[kc] = cell_lo[ncell];
    pl.kcells.step_lo[kc] = ks;
    pl.kcells.nkp = ks;
  }
  pl.nvt = (p + pl.TVX - 1) / pl.TVX;
  pl.nsplit = (int)std::min<int64_t>(items, std::max<int64_t>(1, (1024 + pl.nvt - 1) / pl.nvt));
  // the row-offset table of a workgroup's items sits in LDS next to the X tile:
  // split the items further until it fits in 16 KiB (or what the tile leaves)
  {
    const size_t room = std::min<size_t>(16 * 1024, 160 * 1024 - std::min<size_t>(lds_x, 160 * 1024));
    const size_t per_item = (size_t)steps * 16;
    if (room < per_item + 128) return false;
    const int64_t fit = (int64_t)((room - 128) / per_item);
    pl.nsplit = (int)std::max<int64_t>(pl.nsplit, (items + fit - 1) / fit);
    const int64_t per = (items + pl.nsplit - 1) / pl.nsplit;
    pl.lds = lds_x + ((size_t)per * steps + 8) * 16;
  }
  if (pl.lds > 160 * 1024) return false;
  pl.nslab = pl.nvt * pl.VB;
  pl.nchunk = (int)((pl.nslab + 63) / 64);
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t o = off;
    off += (bytes + 255) / 256 * 256;
    return o;
  };
  const size_t E = (size_t)items * pl.MC * 16;
  pl.o_frag = take(((size_t)pl.MC * items * steps + 8) * 64 * sizeof(double));
  pl.o_rowoff = take(((size_t)items * steps + 8) * 4 * sizeof(int32_t));
  pl.o_sc = take((size_t)items * ncell * p * sizeof(double));
  pl.o_sh = take((size_t)items * ncell * p * sizeof(double));
  pl.nslabm = pl.flat ? 2 * pl.nsplit : pl.nsplit;
  pl.o_mom = take(moments ? (size_t)2 * pl.nslabm * p * k * sizeof(double) : 0);
  pl.o_sq = take(rowsq ? (size_t)pl.nslab * E * sizeof(double) : 0);
  pl.o_sq2 = take(rowsq ? (size_t)pl.nchunk * E * sizeof(double) : 0);
  pl.bytes = off;
  return true;
}

template <int NT, int TVX>
int run_fused(const FusedArgs &a, const FusedPlan &pl, hipStream_t st) {
  auto kern = item_fused2_kernel<NT, TVX>;
  if (pl.lds > 64 * 1024 &&
      hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds) != hipSuccess)
    return PLSR_ELAUNCH;
  hipLaunchKernelGGL(kern, dim3((unsigned)pl.nvt, (unsigned)pl.nsplit), dim3(pl.waves * 64), pl.lds, st, a);
  return launch_ok();
}
}  // namespace

extern "C" size_t plsr_item_fused_workspace_bytes(int32_t n, int32_t nz, int32_t k, const int32_t *cell_lo,
                                                  int32_t ncell, int32_t items, int64_t p,
                                                  int32_t want_moments, int32_t want_rowsq) {
  FusedPlan pl;
  return fused_plan(n, nz, k, cell_lo, nullptr, ncell, items, p, want_moments != 0, want_rowsq != 0, pl)
             ? pl.bytes
             : 0;
}

extern "C" int plsr_item_fused(const double *d_X, int64_t ldx, int64_t p, int32_t n, const int32_t *d_src,
                               int32_t nz, const int32_t *cell_lo, const int32_t *cell_z, int32_t ncell,
                               const double *d_rows, int32_t items, int32_t k, const double *d_ref,
                               double *d_S1, double *d_S2, double *d_vst, int64_t ldv, double *d_rowsq,
                               double *d_sc, double *d_sh, int32_t stats_ready, void *d_work,
                               size_t work_bytes, void *stream) {
  if (!d_X || !d_src || !d_rows || !d_work || !cell_lo || !cell_z || ldx < p) return PLSR_EINVAL;
  if ((d_sc == nullptr) != (d_sh == nullptr) || (stats_ready && !d_sc)) return PLSR_EINVAL;
  if ((d_S1 == nullptr) != (d_S2 == nullptr) || (d_vst && ldv < p)) return PLSR_EINVAL;
  FusedPlan pl;
  if (!fused_plan(n, nz, k, cell_lo, cell_z, ncell, items, p, d_S1 != nullptr, d_rowsq != nullptr, pl))
    return PLSR_EUNSUPPORTED;
  if (pl.bytes > work_bytes) return PLSR_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  char *w = (char *)d_work;

  StatsArgs sa;
  sa.X = d_X;
  sa.ldx = ldx;
  sa.p = p;
  sa.n = n;
  sa.nz = nz;
  sa.items = items;
  sa.src = d_src;
  sa.cells = pl.cells;
  sa.sc = d_sc ? d_sc : (double *)(w + pl.o_sc);
  sa.sh = d_sh ? d_sh : (double *)(w + pl.o_sh);
  const size_t lds_stats = (size_t)n * TV * sizeof(double);
  if (lds_stats > 64 * 1024 &&
      hipFuncSetAttribute((const void *)item_stats_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)lds_stats) != hipSuccess)
    return PLSR_ELAUNCH;
  if (!stats_ready)
    hipLaunchKernelGGL(item_stats_kernel, dim3((unsigned)((p + TV - 1) / TV), (unsigned)pl.nsplit),
                       dim3(64 * STATS_WAVES), lds_stats, st, sa);

  MetaArgs ma;
  ma.rows = d_rows;
  ma.src = d_src;
  ma.items = items;
  ma.k = k;
  ma.nz = nz;
  ma.MC = pl.MC;
  ma.row_bytes = pl.TVX * 8;
  ma.cells = pl.kcells;
  ma.frag = (double *)(w + pl.o_frag);
  ma.rowoff = (int32_t *)(w + pl.o_rowoff);
  const int64_t total = (int64_t)pl.MC * items * pl.cells.nkp * 64;
  // the prefetch rings run 8 k-steps past the end: keep that padding defined
  (void)hipMemsetAsync(ma.frag + total, 0, 8 * 64 * sizeof(double), st);
  (void)hipMemsetAsync(ma.rowoff + (size_t)items * pl.cells.nkp * 4, 0, 8 * 4 * sizeof(int32_t), st);
  hipLaunchKernelGGL(item_meta_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, ma);

  FusedArgs a;
  a.X = d_X;
  a.ldx = ldx;
  a.p = p;
  a.n = n;
  a.items = items;
  a.k = k;
  a.MC = pl.MC;
  a.cells = pl.kcells;
  a.nstat = pl.cells.ncell;
  a.frag = ma.frag;
  a.rowoff = ma.rowoff;
  a.sc = sa.sc;
  a.sh = sa.sh;
  a.S1 = d_S1 ? (double *)(w + pl.o_mom) : nullptr;
  a.S2 = d_S1 ? a.S1 + (size_t)pl.nslabm * p * k : nullptr;
  a.flat = pl.flat;
  if (d_S1 && pl.flat)   // a (tile, part) pair without items writes nothing
    (void)hipMemsetAsync(a.S1, 0, (size_t)2 * pl.nslabm * p * k * sizeof(double), st);
  a.vst = d_vst;
  a.ldv = ldv;
  a.rowsq_part = d_rowsq ? (double *)(w + pl.o_sq) : nullptr;
  int rc = PLSR_EUNSUPPORTED;
  if (pl.TVX == 64 && pl.NT == 1) rc = run_fused<1, 64>(a, pl, st);
  if (pl.TVX == 64 && pl.NT == 2) rc = run_fused<2, 64>(a, pl, st);
  if (pl.TVX == 64 && pl.NT == 4) rc = run_fused<4, 64>(a, pl, st);
  if (rc) return rc;
  if (d_S1) {
    const int64_t cnt = p * k;
    dim3 g((unsigned)((cnt + 255) / 256));
    hipLaunchKernelGGL(moment_unshift_kernel, g, dim3(256), 0, st, d_S1, d_S2, (const double *)a.S1,
                       (const double *)a.S2, d_ref, cnt, pl.nslabm, (double)items);
  }
  if (d_rowsq) {
    const int64_t E = (int64_t)items * pl.MC * 16;
    double *lvl2 = (double *)(w + pl.o_sq2);
    hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)((E + 255) / 256), (unsigned)pl.nchunk), dim3(256), 0, st,
                       (const double *)a.rowsq_part, lvl2, E, (int)pl.nslab, 64);
    hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)((E + 255) / 256), 1), dim3(256), 0, st,
                       (const double *)lvl2, d_rowsq, E, pl.nchunk, pl.nchunk);
  }
  return launch_ok();
}

// ---------------------------------------------------------------------------
// K2 with the gather / z-score fused in (split-half of rb / mb)
// ---------------------------------------------------------------------------
namespace {
struct GramFusedPlan {
  GramPlan g;
  FusedCells cells;
  size_t o_sc, o_sh, o_cell, o_act, o_tab, bytes;
};

bool gram_fused_plan(int32_t n, int32_t nz, int32_t m, const int32_t *cell_lo, const int32_t *cell_z,
                     int32_t ncell, int32_t items, int64_t p, GramFusedPlan &pl) {
  if (n <= 0 || !cell_lo || ncell <= 0 || ncell > FZ_MAXCELL || cell_lo[0] != 0 || cell_lo[ncell] != nz)
    return false;
  if ((size_t)n * TV * sizeof(double) > 160 * 1024) return false;        // statistics kernel's X tile
  if (nz > 256) return false;          // (one 64-bit activity mask per tile group: nk <= 64)
  if (p >= ((int64_t)1 << 29)) return false;      // the fused Gram addresses voxels by a 32-bit byte offset
  if (!gram_plan(nz, m, items, p, true, pl.g)) return false;
  pl.cells.ncell = ncell;
  int steps = 0;
  for (int c = 0; c < ncell; ++c) {
    if (cell_lo[c + 1] <= cell_lo[c]) return false;
    pl.cells.row_lo[c] = cell_lo[c];
    pl.cells.step_lo[c] = steps;
    pl.cells.z[c] = cell_z ? cell_z[c] : 1;
    steps += (cell_lo[c + 1] - cell_lo[c] + 3) / 4;
  }
  pl.cells.row_lo[ncell] = nz;
  pl.cells.step_lo[ncell] = steps;
  pl.cells.nkp = steps;
  size_t off = pl.g.bytes;
  auto take = [&](size_t bytes) {
    const size_t o = off;
    off += (bytes + 255) / 256 * 256;
    return o;
  };
  pl.o_sc = take((size_t)items * ncell * p * sizeof(double));
  pl.o_sh = take((size_t)items * ncell * p * sizeof(double));
  pl.o_cell = take((size_t)nz * sizeof(int32_t));
  pl.o_act = take(2 * sizeof(uint64_t));
  pl.o_tab = take((size_t)items * (4 * ((nz + 3) / 4) + GRAM_PF) * 4 * sizeof(int64_t));
  pl.bytes = off;
  return true;
}

__global__ void rowcell_kernel(FusedCells cells, int nz, int32_t *out) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= nz) return;
  int c = 0;
  while (c + 1 < cells.ncell && r >= cells.row_lo[c + 1]) ++c;
  out[r] = c;
}
}  // namespace

extern "C" size_t plsr_gram_fused_workspace_bytes(int32_t n, int32_t nz, int32_t m, const int32_t *cell_lo,
                                                  int32_t ncell, int32_t items, int64_t p) {
  GramFusedPlan pl;
  return gram_fused_plan(n, nz, m, cell_lo, nullptr, ncell, items, p, pl) ? pl.bytes : 0;
}

extern "C" int plsr_gram_fused(const double *d_X, int64_t ldx, int64_t p, int32_t n, const int32_t *d_src,
                               int32_t nz, const int32_t *cell_lo, const int32_t *cell_z, int32_t ncell,
                               const double *d_frag, int32_t items, int32_t m, double *d_G, void *d_work,
                               size_t work_bytes, void *stream) {
  if (!d_X || !d_src || !d_frag || !d_G || !d_work || !cell_lo || !cell_z || ldx < p) return PLSR_EINVAL;
  GramFusedPlan pl;
  if (!gram_fused_plan(n, nz, m, cell_lo, cell_z, ncell, items, p, pl)) return PLSR_EUNSUPPORTED;
  if (pl.bytes > work_bytes) return PLSR_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  char *w = (char *)d_work;

  StatsArgs sa;
  sa.X = d_X;
  sa.ldx = ldx;
  sa.p = p;
  sa.n = n;
  sa.nz = nz;
  sa.items = items;
  sa.src = d_src;
  sa.cells = pl.cells;
  sa.sc = (double *)(w + pl.o_sc);
  sa.sh = (double *)(w + pl.o_sh);
  const size_t lds_stats = (size_t)n * TV * sizeof(double);
  if (lds_stats > 64 * 1024 &&
      hipFuncSetAttribute((const void *)item_stats_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)lds_stats) != hipSuccess)
    return PLSR_ELAUNCH;
  const int64_t nvt = (p + TV - 1) / TV;
  const int nsplit = (int)std::min<int64_t>(items, std::max<int64_t>(1, (1024 + nvt - 1) / nvt));
  hipLaunchKernelGGL(item_stats_kernel, dim3((unsigned)nvt, (unsigned)nsplit), dim3(64 * STATS_WAVES), lds_stats, st,
                     sa);
  int32_t *rowcell = (int32_t *)(w + pl.o_cell);
  hipLaunchKernelGGL(rowcell_kernel, dim3((unsigned)((nz + 255) / 256)), dim3(256), 0, st, pl.cells, nz, rowcell);

  const GramPlan &g = pl.g;
  GramArgs a;
  a.X = d_X;
  a.x_item_stride = 0;
  a.ldx = ldx;
  a.p = p;
  a.n = nz;
  a.nk = (nz + 3) / 4;
  a.frag = d_frag;
  a.items = items;
  a.tiles_per_chunk = g.tiles_per_chunk;
  a.ks = g.ks;
  a.G_part = (double *)d_work;
  a.src = d_src;
  a.rowcell = rowcell;
  a.ncell = ncell;
  a.sc = sa.sc;
  a.sh = sa.sh;
  {
    int64_t *tab = (int64_t *)(w + pl.o_tab);
    const int ntr = 4 * a.nk + GRAM_PF;
    const int64_t cnt = (int64_t)items * ntr;
    hipLaunchKernelGGL(gram_rowtab_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, d_src,
                       (const int32_t *)rowcell, items, nz, ntr, ncell, d_X, ldx, (const double *)sa.sc,
                       (const double *)sa.sh, p, tab);
    a.rowtab = tab;
  }
  // tile groups for an operator of two halves of m / 2 rows each (the split-half items): the low
  // group ends where the second half's rows begin, the high group begins where the first half's end;
  // what the groups may skip is read off the fragments (gram_activity_kernel), whatever the operator is
  const int half = m / 2;
  const int SL = g.MC > 1 ? half / 16 : 0, SH = g.MC > 1 ? (half + 15) / 16 : g.MC;
  if (g.MC > 1) {
    unsigned long long *act = (unsigned long long *)(w + pl.o_act);
    if (hipMemsetAsync(act, 0, 2 * sizeof(uint64_t), st) != hipSuccess) return PLSR_ELAUNCH;
    const int64_t total = (int64_t)items * g.MC * a.nk * 64;
    const unsigned nb = (unsigned)std::min<int64_t>(2048, (total + 255) / 256);
    hipLaunchKernelGGL(gram_activity_kernel, dim3(nb), dim3(256), 0, st, d_frag, total, g.MC, a.nk, SL, SH, act);
    a.act = (const uint64_t *)act;
  } else {
    a.act = nullptr;
  }
  int rc = PLSR_EUNSUPPORTED;
#define PLSR_GS(MCv, SLv, SHv) \
  if (g.MC == MCv && SL == SLv && SH == SHv) rc = launch_gram<MCv, 1, true, SLv, SHv>(a, g, st);
  if (g.MC == 1) rc = launch_gram<1, 1, true>(a, g, st);
  PLSR_GS(2, 0, 1) PLSR_GS(2, 1, 1) PLSR_GS(3, 1, 1) PLSR_GS(3, 1, 2) PLSR_GS(4, 1, 2) PLSR_GS(4, 2, 2)
  PLSR_GS(5, 2, 2) PLSR_GS(5, 2, 3) PLSR_GS(6, 2, 3) PLSR_GS(6, 3, 3)
#undef PLSR_GS
  if (rc) return rc;
  const int64_t E = (int64_t)items * g.MM * g.MM;
  hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)((E + 255) / 256), 1), dim3(256), 0, st,
                     (const double *)d_work, d_G, E, g.nchunk, g.nchunk);
  return launch_ok();
}

extern "C" int plsr_scale_project_rows(const double *d_raw, const double *d_rowsq, int64_t rowsq_stride,
                                       const double *d_U, int32_t items, int32_t kr, int32_t nz, int32_t k,
                                       double *d_out, void *stream) {
  if (!d_raw || !d_rowsq || !d_U || !d_out || items <= 0 || kr <= 0 || nz <= 0 || k <= 0 || rowsq_stride < kr)
    return PLSR_EINVAL;
  const int64_t total = (int64_t)items * k * nz;
  hipLaunchKernelGGL(scale_project_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, d_raw, d_rowsq, rowsq_stride, d_U, items, kr, nz, k, d_out);
  return launch_ok();
}

// ---------------------------------------------------------------------------
// K4a: the same product on aggregated operators, X in registers (plsr_agg.hip.h)
// ---------------------------------------------------------------------------
namespace {
struct AggPlan {
  AggProgram prog;
  int MC, NF, nsplit, per, ngroups, unit, nups, us;
  int64_t nwg, ntile;
  int nchunk;
  size_t lds;
  size_t o_prog, o_afrag, o_mfrag, o_mom, o_sq, o_sq2, bytes;
};

// fragment counts item_agg_kernel is instantiated for
int agg_nf_instance(int nf) {
  static const int inst[] = {8, 16, 24, 30, 32};
  for (int v : inst)
    if (nf <= v) return v;
  return 0;
}

// 32-bit lane offsets of the result stores of K4a / K4b (rows p apart; the calls check their actual stride)
inline bool item_store_offsets_fit(int64_t p) { return (12 * p + p) * 8 < ((int64_t)1 << 32); }

bool agg_plan(int32_t n, int32_t nz, int32_t k, const int32_t *cell_lo, const int32_t *cell_z,
              const int32_t *src_lo, const int32_t *src_hi, int32_t ncell, int32_t items, int64_t p,
              bool moments, bool rowsq, AggPlan &pl) {
  if (n <= 0 || n > AG_MAXROWS || nz <= 0 || k <= 0 || items <= 0 || p <= 0) return false;
  if (!item_store_offsets_fit(p)) return false;
  if (!cell_lo || !cell_z || !src_lo || !src_hi || ncell <= 0 || ncell > FZ_MAXCELL) return false;
  if (cell_lo[0] != 0 || cell_lo[ncell] != nz) return false;
  pl.MC = (k + 15) / 16;
  if (pl.MC > 3) return false;
  pl.NF = agg_nf_instance((n + 3) / 4);
  if (!pl.NF) return false;
  AggProgram &g = pl.prog;
  memset(&g, 0, sizeof(g));
  memset(g.rowcell, 0xff, sizeof(g.rowcell));
  // cells to sweeps: ascending first fragment, each into the first sweep whose cells end before it
  int order[FZ_MAXCELL], last[AG_MAXSWEEP], sweep_of[FZ_MAXCELL];
  int old_steps = 0;
  for (int c = 0; c < ncell; ++c) {
    if (cell_lo[c + 1] <= cell_lo[c] || src_lo[c] < 0 || src_hi[c] > n || src_hi[c] <= src_lo[c]) return false;
    order[c] = c;
    old_steps += (cell_lo[c + 1] - cell_lo[c] + 3) / 4;
  }
  std::stable_sort(order, order + ncell, [&](int a, int b) { return src_lo[a] < src_lo[b]; });
  g.nsweep = 0;
  for (int q = 0; q < ncell; ++q) {
    const int c = order[q];
    const int f0 = src_lo[c] / 4, f1 = (src_hi[c] - 1) / 4;
    int sw = 0;
    while (sw < g.nsweep && last[sw] >= f0) ++sw;
    if (sw == g.nsweep) {
      if (g.nsweep == AG_MAXSWEEP) return false;
      ++g.nsweep;
    }
    last[sw] = f1;
    sweep_of[c] = sw;
    g.start[sw] |= 1u << f0;
    if (cell_z[c]) g.zstart[sw] |= 1u << f0;
    for (int r = src_lo[c]; r < src_hi[c]; ++r) g.rowcell[sw][r] = (int8_t)c;
  }
  // dense sweeps: every sweep walks all NF fragments -- not worth it when the cells leave most
  // of them empty (arbitrary gathers: every cell may read every row)
  if ((int64_t)g.nsweep * pl.NF * 3 > (int64_t)old_steps * 4 + 3 * 8) return false;
  // z-scored cells in the order the kernel meets them: sweep by sweep, ascending fragment
  g.nzsweep = 0;
  g.nzc = 0;
  for (int sw = 0; sw < g.nsweep; ++sw) {
    if (!g.zstart[sw]) continue;
    const int zs = g.nzsweep++;
    g.zsweep[zs] = sw;
    for (int q = 0; q < ncell; ++q) {
      const int c = order[q];
      if (sweep_of[c] != sw || !cell_z[c]) continue;
      if (g.nzc == AG_MAXZC) return false;
      g.zend[zs] |= 1u << ((src_hi[c] - 1) / 4);
      g.zc_zs[g.nzc] = (int8_t)zs;
      g.zc_flo[g.nzc] = (int8_t)(src_lo[c] / 4);
      g.zc_fhi[g.nzc] = (int8_t)((src_hi[c] - 1) / 4);
      g.cnt[g.nzc] = (double)(cell_lo[c + 1] - cell_lo[c]);
      g.rcnt[g.nzc] = 1.0 / g.cnt[g.nzc];
      ++g.nzc;
    }
  }
  pl.ngroups = (items + 3) / 4;
  pl.nwg = (p + 16 * AG_WAVES - 1) / (16 * AG_WAVES);
  pl.ntile = pl.nwg * AG_WAVES;
  // splits of the items: fill the 512 workgroup slots of the chip (two per CU) evenly; every
  // split costs a slab of partial moment sums, so take the first that wastes < 6 % of the last round
  pl.nsplit = 1;
  {
    double best = 0.0;
    for (int ns = 1; ns <= std::min(8, pl.ngroups); ++ns) {
      const double rounds = (double)pl.nwg * ns / 512.0;
      const double eff = rounds / std::ceil(rounds);
      if (eff > best + 1e-9) {
        best = eff;
        pl.nsplit = ns;
      }
      if (eff >= 0.94) break;
    }
  }
  pl.per = (pl.ngroups + pl.nsplit - 1) / pl.nsplit * 4;
  pl.nsplit = (items + pl.per - 1) / pl.per;
  pl.unit = agg_unit_elems(pl.NF, pl.MC);
  pl.nups = agg_units_per_sweep(pl.NF, pl.MC);
  pl.us = agg_unit_steps(pl.NF, pl.MC);
  pl.lds = ((size_t)2 * pl.unit + (size_t)AG_WAVES * g.nzc * 128 + 2 * AG_MAXZC) * sizeof(double);   // two units, scale / shift, counts
  if (pl.lds > 160 * 1024) return false;
  pl.nchunk = (int)((pl.ntile + 63) / 64);
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t o = off;
    off += (bytes + 255) / 256 * 256;
    return o;
  };
  const size_t E = (size_t)items * pl.MC * 16;
  pl.o_prog = take(sizeof(AggProgram));
  pl.o_afrag = take(((size_t)items * g.nsweep * pl.nups + 1) * pl.unit * sizeof(double));   // + the unit the staging reads ahead
  pl.o_mfrag = take(((size_t)pl.ngroups * g.nzsweep * pl.NF + AG_SRING) * 64 * sizeof(double));
  pl.o_mom = take(moments ? (size_t)2 * pl.nsplit * p * k * sizeof(double) : 0);
  pl.o_sq = take(rowsq ? (size_t)pl.ntile * E * sizeof(double) : 0);
  pl.o_sq2 = take(rowsq ? (size_t)pl.nchunk * E * sizeof(double) : 0);
  pl.bytes = off;
  return true;
}

template <int NF, int MC>
int run_agg(const AggArgs &a, const AggPlan &pl, hipStream_t st) {
  auto kern = item_agg_kernel<NF, MC>;
  if (pl.lds > 64 * 1024 &&
      hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds) != hipSuccess)
    return PLSR_ELAUNCH;
  hipLaunchKernelGGL(kern, dim3((unsigned)pl.nwg, (unsigned)pl.nsplit), dim3(AG_WAVES * 64), pl.lds, st, a);
  return launch_ok();
}

template <int NF>
int run_agg_mc(const AggArgs &a, const AggPlan &pl, hipStream_t st) {
  switch (pl.MC) {
    case 1: return run_agg<NF, 1>(a, pl, st);
    case 2: return run_agg<NF, 2>(a, pl, st);
    case 3: return run_agg<NF, 3>(a, pl, st);
  }
  return PLSR_EUNSUPPORTED;
}
}  // namespace

extern "C" size_t plsr_item_agg_workspace_bytes(int32_t n, int32_t nz, int32_t k, const int32_t *cell_lo,
                                                const int32_t *cell_z, const int32_t *src_lo,
                                                const int32_t *src_hi, int32_t ncell, int32_t items, int64_t p,
                                                int32_t want_moments, int32_t want_rowsq) {
  AggPlan pl;
  return agg_plan(n, nz, k, cell_lo, cell_z, src_lo, src_hi, ncell, items, p, want_moments != 0, want_rowsq != 0,
                  pl)
             ? pl.bytes
             : 0;
}

extern "C" int plsr_item_agg(const double *d_X, int64_t ldx, int64_t p, int32_t n, const int32_t *d_src, int32_t nz,
                             const int32_t *cell_lo, const int32_t *cell_z, const int32_t *src_lo,
                             const int32_t *src_hi, int32_t ncell, const double *d_rows, int32_t items, int32_t k,
                             const double *d_ref, double *d_S1, double *d_S2, double *d_vst, int64_t ldv,
                             double *d_rowsq, void *d_work, size_t work_bytes, void *stream) {
  if (!d_X || !d_src || !d_rows || !d_work || !cell_lo || !cell_z || !src_lo || !src_hi || ldx < p)
    return PLSR_EINVAL;
  if ((d_S1 == nullptr) != (d_S2 == nullptr) || (d_vst && ldv < p)) return PLSR_EINVAL;
  if (d_vst && (12 * ldv + p) * 8 >= ((int64_t)1 << 32)) return PLSR_EUNSUPPORTED;   // 32-bit lane offsets of the stores
  AggPlan pl;
  if (!agg_plan(n, nz, k, cell_lo, cell_z, src_lo, src_hi, ncell, items, p, d_S1 != nullptr, d_rowsq != nullptr, pl))
    return PLSR_EUNSUPPORTED;
  if (pl.bytes > work_bytes) return PLSR_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  char *w = (char *)d_work;

  AggMetaArgs ma;
  ma.rows = d_rows;
  ma.src = d_src;
  ma.items = items;
  ma.k = k;
  ma.nz = nz;
  ma.MC = pl.MC;
  ma.NF = pl.NF;
  ma.ncell = ncell;
  for (int c = 0; c < ncell; ++c) {
    ma.cell_lo[c] = cell_lo[c];
    ma.src_lo[c] = src_lo[c];
    ma.src_hi[c] = src_hi[c];
    ma.cell_z[c] = cell_z[c];
  }
  ma.cell_lo[ncell] = cell_lo[ncell];
  ma.prog = pl.prog;
  ma.afrag = (double *)(w + pl.o_afrag);
  ma.mfrag = (double *)(w + pl.o_mfrag);
  ma.prog_out = (AggProgram *)(w + pl.o_prog);
  ma.unit = pl.unit;
  ma.nups = pl.nups;
  ma.us = pl.us;
  const int64_t na = (int64_t)items * pl.prog.nsweep * pl.nups * pl.unit;
  const int64_t nm = (int64_t)pl.ngroups * pl.prog.nzsweep * pl.NF * 64;
  // the multiplicity ring runs AG_SRING k-steps past the end: keep that padding defined
  (void)hipMemsetAsync(ma.mfrag + nm, 0, (size_t)AG_SRING * 64 * sizeof(double), st);
  hipLaunchKernelGGL(agg_meta_kernel, dim3((unsigned)((na + nm + 255) / 256)), dim3(256), 0, st, ma);
  hipLaunchKernelGGL(agg_check_kernel, dim3((unsigned)(((int64_t)items * nz + 255) / 256)), dim3(256), 0, st, ma);

  AggArgs a;
  a.X = d_X;
  a.ldx = ldx;
  a.p = p;
  a.n = n;
  a.items = items;
  a.k = k;
  a.per = pl.per;
  a.prog = ma.prog_out;
  a.afrag = ma.afrag;
  a.mfrag = ma.mfrag;
  a.S1 = d_S1 ? (double *)(w + pl.o_mom) : nullptr;
  a.S2 = d_S1 ? a.S1 + (size_t)pl.nsplit * p * k : nullptr;
  a.vst = d_vst;
  a.ldv = ldv;
  a.rowsq_part = d_rowsq ? (double *)(w + pl.o_sq) : nullptr;
#ifdef AGG_TIMING
  static long long *dbg = nullptr;
  const size_t ndbg = (size_t)pl.nwg * pl.nsplit * AG_WAVES * 8;
  if (!dbg) (void)hipMalloc(&dbg, 64 << 20);
  (void)hipMemsetAsync(dbg, 0, ndbg * 8, st);
  a.dbg = dbg;
#endif
  int rc = PLSR_EUNSUPPORTED;
  switch (pl.NF) {
    case 8: rc = run_agg_mc<8>(a, pl, st); break;
    case 16: rc = run_agg_mc<16>(a, pl, st); break;
    case 24: rc = run_agg_mc<24>(a, pl, st); break;
    case 30: rc = run_agg_mc<30>(a, pl, st); break;
    case 32: rc = run_agg_mc<32>(a, pl, st); break;
  }
  if (rc) return rc;
#ifdef AGG_TIMING
  {
    std::vector<long long> h(ndbg);
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(h.data(), dbg, ndbg * 8, hipMemcpyDeviceToHost);
    double sum[5] = {0, 0, 0, 0, 0};
    long long tmin = -1, tmax = 0, n_it = 0;
    for (size_t i = 0; i < ndbg / 8; ++i) {
      for (int q = 0; q < 5; ++q) sum[q] += (double)h[i * 8 + q];
      n_it += h[i * 8 + 5];
      if (tmin < 0 || h[i * 8 + 6] < tmin) tmin = h[i * 8 + 6];
      tmax = std::max(tmax, h[i * 8 + 6] + h[i * 8]);
    }
    fprintf(stderr, "[agg timing] waves %zu, clock ticks per wave-item: total %.0f stats %.0f loop %.0f (barrier %.0f) "
            "epilogue %.0f; kernel span %lld ticks\n", ndbg / 8, sum[0] / n_it, sum[1] / n_it, sum[2] / n_it,
            sum[3] / n_it, sum[4] / n_it, tmax - tmin);
  }
#endif
  if (d_S1) {
    const int64_t cnt = p * k;
    hipLaunchKernelGGL(moment_unshift_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, d_S1, d_S2,
                       (const double *)a.S1, (const double *)a.S2, d_ref, cnt, pl.nsplit, (double)items);
  }
  if (d_rowsq) {
    const int64_t E = (int64_t)items * pl.MC * 16;
    double *lvl2 = (double *)(w + pl.o_sq2);
    hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)((E + 255) / 256), (unsigned)pl.nchunk), dim3(256), 0, st,
                       (const double *)a.rowsq_part, lvl2, E, (int)pl.ntile, 64);
    hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)((E + 255) / 256), 1), dim3(256), 0, st,
                       (const double *)lvl2, d_rowsq, E, pl.nchunk, pl.nchunk);
  }
  return launch_ok();
}

// ---------------------------------------------------------------------------
// K4b: behaviour PLS bootstrap in two stages (plsr_beh.hip.h)
// ---------------------------------------------------------------------------
namespace {
struct BehPlan {
  int CSMAX, NCMAX, BP, IP, MC, cs, nsplit, per, ngrp, nsg, unit, unit_p;
  int64_t nwg;
  size_t lds;
  size_t o_a1, o_mfrag, o_u2, o_mom, bytes;
};

// developer builds (-DPLSR_DEV_KNOBS): PLSR_BEH_GENERIC selects the guarded instance for A/B measurements
inline bool beh_force_generic() {
#ifdef PLSR_DEV_KNOBS
  return getenv("PLSR_BEH_GENERIC") != nullptr;
#else
  return false;
#endif
}

bool beh_plan(int32_t n, int32_t nz, int32_t b, int32_t k, const int32_t *cell_lo, const int32_t *src_lo,
              const int32_t *src_hi, int32_t ncell, int32_t items, int64_t p, bool moments, BehPlan &pl) {
  if (n <= 0 || nz <= 0 || b <= 0 || b > 16 || k <= 0 || k > 48 || items <= 0 || p <= 0) return false;
  if (!item_store_offsets_fit(p)) return false;
  if (!cell_lo || !src_lo || !src_hi || ncell <= 0 || ncell > AG_MAXZC) return false;
  if (cell_lo[0] != 0 || cell_lo[ncell] != nz) return false;
  int cs = 0;
  for (int c = 0; c < ncell; ++c) {
    if (cell_lo[c + 1] <= cell_lo[c] || src_lo[c] < 0 || src_hi[c] > n || src_hi[c] <= src_lo[c]) return false;
    cs = std::max(cs, (src_hi[c] - src_lo[c] + 3) / 4);
  }
  // register layouts item_beh_kernel is instantiated for: (k-steps per cell, cells)
  static const int lay[][2] = {{5, 6}, {8, 4}, {4, 8}, {2, 16}};
  pl.CSMAX = 0;
  for (auto &l : lay)
    if (cs <= l[0] && ncell <= l[1]) {
      pl.CSMAX = l[0];
      pl.NCMAX = l[1];
      break;
    }
  if (!pl.CSMAX) return false;
  pl.cs = cs;
  pl.BP = b <= 8 ? 8 : 16;
  pl.IP = 16 / pl.BP;
  pl.MC = (k + 15) / 16;
  pl.ngrp = (items + pl.IP - 1) / pl.IP;
  pl.nsg = (items + 3) / 4;
  pl.unit = ncell * cs * 64;
  pl.unit_p = (pl.unit + 511) / 512 * 512;
  pl.nwg = (p + 16 * BH_WAVES - 1) / (16 * BH_WAVES);
  pl.nsplit = 1;
  {
    double best = 0.0;
    for (int ns = 1; ns <= std::min(8, pl.nsg); ++ns) {
      const double rounds = (double)pl.nwg * ns / 512.0;
      const double eff = rounds / std::ceil(rounds);
      if (eff > best + 1e-9) {
        best = eff;
        pl.nsplit = ns;
      }
      if (eff >= 0.94) break;
    }
  }
  pl.per = (pl.nsg + pl.nsplit - 1) / pl.nsplit * 4;
  pl.nsplit = (items + pl.per - 1) / pl.per;
  pl.lds = ((size_t)ncell * (pl.BP / 4) * pl.MC * 64 + (size_t)2 * pl.unit_p + (size_t)BH_WAVES * ncell * 64 +
            2 * AG_MAXZC) * sizeof(double);
  if (pl.lds > 160 * 1024) return false;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t o = off;
    off += (bytes + 255) / 256 * 256;
    return o;
  };
  pl.o_a1 = take(((size_t)pl.ngrp * pl.unit + 2 * pl.unit_p) * sizeof(double));   // + what the staging reads ahead
  pl.o_mfrag = take((size_t)pl.nsg * ncell * cs * 64 * sizeof(double));
  pl.o_u2 = take((size_t)ncell * (pl.BP / 4) * pl.MC * 64 * sizeof(double));
  pl.o_mom = take(moments ? (size_t)2 * pl.nsplit * p * k * sizeof(double) : 0);
  pl.bytes = off;
  return true;
}

template <int CSMAX, int NCMAX>
int run_beh(const BehArgs &a, const BehPlan &pl, hipStream_t st) {
  auto launch = [&](auto kern) {
    if (pl.lds > 64 * 1024 &&
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds) != hipSuccess)
      return (int)PLSR_ELAUNCH;
    hipLaunchKernelGGL(kern, dim3((unsigned)pl.nwg, (unsigned)pl.nsplit), dim3(BH_WAVES * 64), pl.lds, st, a);
    return launch_ok();
  };
  if (CSMAX == 5 && NCMAX == 6 && pl.BP == 8 && a.ncell == NCMAX && a.cs == CSMAX && a.MC == 3 && !beh_force_generic())
    return launch(item_beh_kernel<5, 6, 8, true>);          // every step live: the instance without guards
  return pl.BP == 8 ? launch(item_beh_kernel<CSMAX, NCMAX, 8>) : launch(item_beh_kernel<CSMAX, NCMAX, 16>);
}
}  // namespace

extern "C" size_t plsr_item_beh_workspace_bytes(int32_t n, int32_t nz, int32_t b, int32_t k, const int32_t *cell_lo,
                                                const int32_t *src_lo, const int32_t *src_hi, int32_t ncell,
                                                int32_t items, int64_t p, int32_t want_moments) {
  BehPlan pl;
  return beh_plan(n, nz, b, k, cell_lo, src_lo, src_hi, ncell, items, p, want_moments != 0, pl) ? pl.bytes : 0;
}

extern "C" int plsr_item_beh(const double *d_X, int64_t ldx, int64_t p, int32_t n, const int32_t *d_src, int32_t nz,
                             const int32_t *cell_lo, const int32_t *src_lo, const int32_t *src_hi, int32_t ncell,
                             const double *d_Yz, int32_t b, const double *d_U, int32_t items, int32_t k,
                             const double *d_ref, double *d_S1, double *d_S2, double *d_vst, int64_t ldv,
                             int32_t vst_tiled, void *d_work, size_t work_bytes, void *stream) {
  if (!d_X || !d_src || !d_Yz || !d_U || !d_work || !cell_lo || !src_lo || !src_hi || ldx < p) return PLSR_EINVAL;
  if ((d_S1 == nullptr) != (d_S2 == nullptr) || (d_vst && ldv < p)) return PLSR_EINVAL;
  if (d_vst && vst_tiled && ldv % LV_T != 0) return PLSR_EINVAL;
  if (d_vst && (12 * ldv + p) * 8 >= ((int64_t)1 << 32)) return PLSR_EUNSUPPORTED;   // 32-bit lane offsets of the stores
  BehPlan pl;
  if (!beh_plan(n, nz, b, k, cell_lo, src_lo, src_hi, ncell, items, p, d_S1 != nullptr, pl)) return PLSR_EUNSUPPORTED;
  if (pl.bytes > work_bytes) return PLSR_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  char *w = (char *)d_work;

  BehMetaArgs ma;
  ma.Yz = d_Yz;
  ma.U = d_U;
  ma.src = d_src;
  ma.items = items;
  ma.nz = nz;
  ma.b = b;
  ma.k = k;
  ma.ncell = ncell;
  ma.cs = pl.cs;
  ma.CSMAX = pl.CSMAX;
  ma.BP = pl.BP;
  ma.IP = pl.IP;
  ma.MC = pl.MC;
  for (int c = 0; c < ncell; ++c) {
    ma.cell_lo[c] = cell_lo[c];
    ma.src_lo[c] = src_lo[c];
    ma.src_hi[c] = src_hi[c];
  }
  ma.cell_lo[ncell] = cell_lo[ncell];
  ma.a1 = (double *)(w + pl.o_a1);
  ma.mfrag = (double *)(w + pl.o_mfrag);
  ma.u2 = (double *)(w + pl.o_u2);
  const int64_t total = (int64_t)pl.ngrp * pl.unit + (int64_t)pl.nsg * ncell * pl.cs * 64 +
                        (int64_t)ncell * (pl.BP / 4) * pl.MC * 64;
  hipLaunchKernelGGL(beh_meta_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, ma);
  hipLaunchKernelGGL(beh_check_kernel, dim3((unsigned)(((int64_t)items * nz + 255) / 256)), dim3(256), 0, st, ma);

  BehArgs a;
  a.X = d_X;
  a.ldx = ldx;
  a.p = p;
  a.n = n;
  a.items = items;
  a.k = k;
  a.per = pl.per;
  a.ncell = ncell;
  a.cs = pl.cs;
  a.MC = pl.MC;
  for (int c = 0; c < ncell; ++c) {
    a.src_lo[c] = src_lo[c];
    a.src_hi[c] = src_hi[c];
    a.cnt[c] = (double)(cell_lo[c + 1] - cell_lo[c]);
    a.rcnt[c] = 1.0 / a.cnt[c];
  }
  a.a1 = ma.a1;
  a.mfrag = ma.mfrag;
  a.u2 = ma.u2;
  a.S1 = d_S1 ? (double *)(w + pl.o_mom) : nullptr;
  a.S2 = d_S1 ? a.S1 + (size_t)pl.nsplit * p * k : nullptr;
  a.vst = d_vst;
  a.ldv = ldv;
  a.vst_tiled = vst_tiled != 0;
#ifdef BEH_TIMING
  static long long *bdbg = nullptr;
  const size_t nbdbg = (size_t)pl.nwg * pl.nsplit * BH_WAVES * 8;
  if (!bdbg) (void)hipMalloc(&bdbg, 64 << 20);
  (void)hipMemsetAsync(bdbg, 0, nbdbg * 8, st);
  a.dbg = bdbg;
#endif
  int rc = PLSR_EUNSUPPORTED;
  if (pl.CSMAX == 5) rc = run_beh<5, 6>(a, pl, st);
  if (pl.CSMAX == 8) rc = run_beh<8, 4>(a, pl, st);
  if (pl.CSMAX == 4) rc = run_beh<4, 8>(a, pl, st);
  if (pl.CSMAX == 2) rc = run_beh<2, 16>(a, pl, st);
  if (rc) return rc;
#ifdef BEH_TIMING
  {
    std::vector<long long> h(nbdbg);
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(h.data(), bdbg, nbdbg * 8, hipMemcpyDeviceToHost);
    double sum[5] = {0, 0, 0, 0, 0};
    long long n_it = 0;
    for (size_t i = 0; i < nbdbg / 8; ++i) {
      for (int q = 0; q < 5; ++q) sum[q] += (double)h[i * 8 + q];
      n_it += h[i * 8 + 5];
    }
    fprintf(stderr, "[beh timing] ticks per wave-item: total %.0f stats %.0f stage1+2 %.0f write+barrier %.0f epilogue %.0f\n",
            sum[0] / n_it, sum[1] / n_it, sum[2] / n_it, sum[3] / n_it, sum[4] / n_it);
  }
#endif
  if (d_S1) {
    const int64_t cnt = p * k;
    hipLaunchKernelGGL(moment_unshift_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, d_S1, d_S2,
                       (const double *)a.S1, (const double *)a.S2, d_ref, cnt, pl.nsplit, (double)items);
  }
  return launch_ok();
}

// ---------------------------------------------------------------------------
// F4: X built on the device (plsr_io.hip.h)
// ---------------------------------------------------------------------------
extern "C" size_t plsr_mask_indices_workspace_bytes(int64_t nvox) {
  if (nvox <= 0) return 0;
  return (size_t)((nvox + IO_BLOCK - 1) / IO_BLOCK + 1) * sizeof(int64_t);
}

extern "C" int plsr_mask_indices(const uint8_t *d_mask, int64_t nvox, int64_t *d_idx, int64_t *d_count, void *d_work,
                                 size_t work_bytes, void *stream) {
  if (!d_mask || !d_idx || !d_count || !d_work || nvox <= 0) return PLSR_EINVAL;
  if (work_bytes < plsr_mask_indices_workspace_bytes(nvox)) return PLSR_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int64_t nb = (nvox + IO_BLOCK - 1) / IO_BLOCK;
  int64_t *cnt = (int64_t *)d_work;
  hipLaunchKernelGGL(mask_count_kernel, dim3((unsigned)nb), dim3(IO_THREADS), 0, st, d_mask, nvox, cnt);
  hipLaunchKernelGGL(mask_scan_kernel, dim3(1), dim3(IO_THREADS), 0, st, cnt, nb, d_count);
  hipLaunchKernelGGL(mask_scatter_kernel, dim3((unsigned)nb), dim3(IO_THREADS), 0, st, d_mask, nvox,
                     (const int64_t *)cnt, d_idx);
  return launch_ok();
}

extern "C" int plsr_mask_apply_rows(const void *d_in, int32_t in_is_f32, int64_t ld_in, int64_t nrows,
                                    const int64_t *d_idx, int64_t nsel, double *d_out, int64_t ld_out, void *stream) {
  if (nrows < 0 || nsel < 0 || ld_out < nsel || nrows > 65535) return PLSR_EINVAL;
  if (nrows == 0 || nsel == 0) return PLSR_OK;                 // (an empty mask: nothing to move)
  if (!d_in || !d_idx || !d_out) return PLSR_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)((nsel + IO_THREADS - 1) / IO_THREADS), (unsigned)nrows);
  if (in_is_f32)
    hipLaunchKernelGGL(mask_apply_kernel<float>, grid, dim3(IO_THREADS), 0, st, (const float *)d_in, ld_in, d_idx, nsel,
                       d_out, ld_out);
  else
    hipLaunchKernelGGL(mask_apply_kernel<double>, grid, dim3(IO_THREADS), 0, st, (const double *)d_in, ld_in, d_idx,
                       nsel, d_out, ld_out);
  return launch_ok();
}
