// K5: latent scores of the rb / mb bootstrap (K4f, plsr_fused.hip.h, produces VS^T).
//
// K5  latent_kernel<MC, NI, IG, WV>   (item-major)
//     Zt_b[j, i] = sum_v VS_b[j, v] * X[i, v]                (k x n per item)
//     nsq_b[j]   = sum_v VS_b[j, v]^2
//     i.e. X @ VS_b (class_functions.py:165-182 as used at
//     bootstrap_permutation.py:638, :647, :655 before the column
//     normalisation) and the norms of :623.  One workgroup owns (group of items,
//     voxel chunk) and accumulates over its voxels in registers; chunk partials
//     are summed afterwards in fixed order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "plsr_project.hip.h"

namespace plsr {

struct LatentArgs {
  const double *X;        // [n][ldx] raw data (shared by all items)
  int64_t ldx, p;
  int32_t n, k, items;
  const double *vst;      // [items][k][ldv]
  int64_t ldv;
  int32_t tiles_per_chunk;   // 32-voxel tiles per voxel chunk
  double *Zt_part;        // [nchunk][items][k][n]
  double *nsq_part;       // [nchunk][items][k]
};

constexpr int LV_T = 32;      // voxels per staged tile
constexpr int XT_VLD = 34;    // K5x's padded LDS row (see latent_xt_kernel)
constexpr int LV_LD = 33;     // padded LDS row: SQ_LDS_BANK_CONFLICT is 0 with 33; with 34 it was 45 % of the LDS cycles

// MC = 16-row tiles of latent variables; NI = 16-row tiles of data rows per wave;
// IG = items per workgroup: they share every staged X tile (X is re-read once
// per IG items instead of once per item) and the two barriers per tile;
// WV = waves per workgroup (each owns NI tiles of data rows).
// The global loads of tile t+1 are issued before the MFMAs of tile t and parked
// in registers, so only the LDS write sits between the two barriers.
template <int MC, int NI, int IG, int WV>
__global__ __launch_bounds__(WV * 64, 2) void latent_kernel(LatentArgs A) {
  const bool want_nsq = A.nsq_part != nullptr;     // (uniform; the column norms usually come from K4f already)
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15;
  const int g = lane >> 4;
  const int item0 = blockIdx.x * IG;
  const int chunk = blockIdx.y;
  constexpr int nrow_x = WV * NI * 16;            // every wave's NI tiles of data rows (zero padded)
  constexpr int RPP = WV * 4;                     // rows staged per pass (sixteen threads per row of 32 voxels)
  constexpr int NV = (IG * MC * 16 + RPP - 1) / RPP;   // passes over the VS^T rows
  constexpr int NX = nrow_x / RPP;                     // passes over the X rows
  double *Vs = smem;                              // [IG][MC*16][LV_LD]   VS^T tiles
  double *Xs = smem + (size_t)IG * MC * 16 * LV_LD;    // [nrow_x][LV_LD]  X tile
  const int64_t nvt = (A.p + LV_T - 1) / LV_T;
  const int64_t t_lo = (int64_t)chunk * A.tiles_per_chunk;
  const int64_t t_hi = min(nvt, t_lo + A.tiles_per_chunk);
  // staging: a thread moves two neighbouring voxels of a row per load (16 bytes); a pass of the
  // workgroup covers RPP rows of the 32-voxel tile
  typedef double d2 __attribute__((ext_vector_type(2)));
  const int srow = tid >> 4, svox = (tid & 15) * 2;

  f64x4 acc[IG][MC][NI];
  double nsq[IG][MC];
#pragma unroll
  for (int ig = 0; ig < IG; ++ig)
#pragma unroll
    for (int mc = 0; mc < MC; ++mc) {
      nsq[ig][mc] = 0.0;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[ig][mc][ni] = (f64x4){0.0, 0.0, 0.0, 0.0};
    }

  // Byte offsets of the rows this thread stages, relative to the first item of the group (VS^T)
  // and to X: constant over the tiles, 32 bits (the library checks the sizes), so that a load is
  // a scalar base + this register and the tile loop carries no 64-bit address arithmetic (the
  // kernel used to spend five vector instructions per MFMA on it).  Rows that do not exist --
  // latent variables past k, data rows past n, items past the last -- are clamped to rows that
  // do: what they produce lands in outputs that are never stored.
  uint32_t offv[NV], offx[NX];
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    const int rr = min(q * RPP + srow, IG * MC * 16 - 1);
    const int ig = rr / (MC * 16), row = min(rr % (MC * 16), A.k - 1);
    const int item = min(item0 + ig, A.items - 1) - item0;
    offv[q] = (uint32_t)((((int64_t)item * A.k + row) * A.ldv + svox) * 8);
  }
#pragma unroll
  for (int q = 0; q < NX; ++q) offx[q] = (uint32_t)(((int64_t)min(q * RPP + srow, A.n - 1) * A.ldx + svox) * 8);
  const char *vbase = (const char *)(A.vst + (int64_t)item0 * A.k * A.ldv);
  const char *xbase = (const char *)A.X;

  d2 pv[NV], px[NX];
  // a tile that is not wholly inside [0, p) and inside the chunk (the last tile of a row, the
  // prefetch past the chunk's end) takes the slow path: per-voxel clamped loads, missing voxels zeroed
  auto fetch = [&](int64_t vt) {
    const bool full = (vt + 1) * LV_T <= A.p && vt < t_hi;        // (uniform)
    if (full) {
      const char *bv = vbase + vt * (LV_T * 8), *bx = xbase + vt * (LV_T * 8);
#pragma unroll
      for (int q = 0; q < NV; ++q) pv[q] = *(const d2 *)(bv + offv[q]);
#pragma unroll
      for (int q = 0; q < NX; ++q) px[q] = *(const d2 *)(bx + offx[q]);
    } else {
      const int64_t v0 = vt * LV_T + svox;
      const bool ok0 = v0 < A.p && vt < t_hi, ok1 = v0 + 1 < A.p && vt < t_hi;
      const int64_t c0 = min(v0, A.p - 1) * 8 - (int64_t)svox * 8, c1 = min(v0 + 1, A.p - 1) * 8 - (int64_t)svox * 8;
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        const double a = *(const double *)(vbase + offv[q] + c0), b = *(const double *)(vbase + offv[q] + c1);
        pv[q] = (d2){ok0 ? a : 0.0, ok1 ? b : 0.0};
      }
#pragma unroll
      for (int q = 0; q < NX; ++q) {
        const double a = *(const double *)(xbase + offx[q] + c0), b = *(const double *)(xbase + offx[q] + c1);
        px[q] = (d2){ok0 ? a : 0.0, ok1 ? b : 0.0};
      }
    }
  };
  auto park = [&]() {
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      const int rr = q * RPP + srow;
      if (rr < IG * MC * 16) {
        Vs[rr * LV_LD + svox] = pv[q].x;
        Vs[rr * LV_LD + svox + 1] = pv[q].y;
      }
    }
#pragma unroll
    for (int q = 0; q < NX; ++q) {
      const int row = q * RPP + srow;
      Xs[row * LV_LD + svox] = px[q].x;
      Xs[row * LV_LD + svox + 1] = px[q].y;
    }
  };

  fetch(t_lo);
  for (int64_t vt = t_lo; vt < t_hi; ++vt) {
    __syncthreads();                  // everybody is done reading the previous tile
    park();
    __syncthreads();
    fetch(vt + 1);                    // in flight during the MFMAs below
    // k-steps of the tile, software-pipelined: the LDS operands of step s + 1 are read into
    // the other of two register sets before the MFMAs of step s are issued
    auto ldk = [&](int s, double (&av)[IG][MC], double (&bv)[NI]) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) bv[ni] = Xs[((wave * NI + ni) * 16 + col) * LV_LD + 4 * s + g];   // B[k = v][n = i]
#pragma unroll
      for (int ig = 0; ig < IG; ++ig)
#pragma unroll
        for (int mc = 0; mc < MC; ++mc) av[ig][mc] = Vs[((ig * MC + mc) * 16 + col) * LV_LD + 4 * s + g];   // A[m = j][k = v]
    };
    auto mmk = [&](double (&av)[IG][MC], double (&bv)[NI]) {
#pragma unroll
      for (int ig = 0; ig < IG; ++ig)
#pragma unroll
        for (int mc = 0; mc < MC; ++mc) {
          if (want_nsq) nsq[ig][mc] = fma(av[ig][mc], av[ig][mc], nsq[ig][mc]);
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) acc[ig][mc][ni] = mfma_f64(av[ig][mc], bv[ni], acc[ig][mc][ni]);
        }
    };
    double a0[IG][MC], a1[IG][MC], b0[NI], b1[NI];
    ldk(0, a0, b0);
#pragma unroll
    for (int s = 0; s < LV_T / 4; s += 2) {
      ldk(s + 1, a1, b1);
      __builtin_amdgcn_sched_barrier(0);
      mmk(a0, b0);
      if (s + 2 < LV_T / 4) ldk(s + 2, a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      mmk(a1, b1);
    }
  }

  // Zt[j = 16 mc + g + 4 r][i = 16 (wave*NI + ni) + col]
#pragma unroll
  for (int ig = 0; ig < IG; ++ig) {
    const int item = item0 + ig;
    if (item >= A.items) break;
    double *zo = A.Zt_part + ((int64_t)chunk * A.items + item) * A.k * A.n;
#pragma unroll
    for (int mc = 0; mc < MC; ++mc)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = mc * 16 + g + 4 * r;
          const int i = (wave * NI + ni) * 16 + col;
          if (j < A.k && i < A.n) zo[(int64_t)j * A.n + i] = acc[ig][mc][ni][r];
        }
    if (wave == 0 && want_nsq) {
#pragma unroll
      for (int mc = 0; mc < MC; ++mc) {
        double x = nsq[ig][mc];
        x += __shfl_xor(x, 16);
        x += __shfl_xor(x, 32);
        const int j = mc * 16 + col;
        if (g == 0 && j < A.k) A.nsq_part[((int64_t)chunk * A.items + item) * A.k + j] = x;
      }
    }
  }
}

inline size_t latent_lds_bytes(int mc, int ni, int ig, int wv) {
  return ((size_t)ig * mc * 16 + (size_t)wv * ni * 16) * LV_LD * sizeof(double);
}

// ---------------------------------------------------------------------------
// K5x: the same product with X read as pre-transposed B fragments (n <= 128).
//
// K5 stages two operands per 32-voxel tile through LDS (the X tile, 128 x 32, is three quarters of
// the bytes) and meets at two barriers per tile.  X^T (voxel-major, rows padded to 128, voxels
// padded with zero rows to whole tiles) is laid out so that the B operand of a k-step -- four voxels
// x the wave's sixteen data rows -- is four contiguous 128-byte segments: every wave loads its own B
// fragments straight into registers, a whole tile ahead, and only VS^T (48 x 32 per tile, shared by
// the waves) goes through LDS, double-buffered, with ONE barrier per tile.  Voxels past p meet zero
// rows of X^T, so the VS^T loads are merely clamped to valid addresses, never zeroed.
// ---------------------------------------------------------------------------
constexpr int XT_LD = 128;     // doubles per voxel row of X^T

__global__ __launch_bounds__(256) void xt_prepare_kernel(const double *X, int64_t ldx, int64_t p, int n, double *XT,
                                                        int64_t p_pad) {
  __shared__ double tile[32][33];
  const int64_t v0 = (int64_t)blockIdx.x * 32;
  const int i0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int i = i0 + r;
    const int64_t v = v0 + tx;
    tile[r][tx] = (i < n && v < p) ? X[(int64_t)i * ldx + v] : 0.0;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int64_t v = v0 + r;
    if (v < p_pad) XT[v * XT_LD + i0 + tx] = tile[tx][r];
  }
}

struct LatentXtArgs {
  const double *XT;       // [p_pad][128]
  int64_t p;
  int32_t n, k, items;
  const double *vst;      // [items][k][ldv]
  int64_t ldv;
  int32_t tiles_per_chunk;
  double *Zt_part;        // [nchunk][items][k][n]
  double *nsq_part;       // [nchunk][items][k] or null
};

// Rows of X an item's sample holds: rows[item][0 .. nu) ascending (padded with the last one), cols[item][i] =
// position of idx[item][i] in that list.  A sample with a row outside [0, n) or with more than max_rows
// different rows gets nu = -1 (latent_index_sum_kernel then fills the item with NaN).  One workgroup of
// XT_LD threads per item.
__global__ __launch_bounds__(128) void latent_index_meta_kernel(const int32_t *idx, int m, int n, int max_rows,
                                                                int32_t *rows, int32_t *nu, int32_t *cols) {
  __shared__ int flag[128];
  __shared__ int rank[128];
  __shared__ int bad, half;
  const int item = blockIdx.x, t = threadIdx.x;
  flag[t] = 0;
  if (t == 0) bad = 0;
  __syncthreads();
  const int32_t *ix = idx + (int64_t)item * m;
  for (int i = t; i < m; i += 128) {
    const int r = ix[i];
    if (r < 0 || r >= n) bad = 1;
    else flag[r] = 1;
  }
  __syncthreads();
  const unsigned long long mask = __ballot(flag[t] != 0);
  if (t == 0) half = __popcll(mask);
  __syncthreads();
  const int below = __popcll(mask & ((1ull << (t & 63)) - 1ull)) + (t >= 64 ? half : 0);
  rank[t] = below;
  __shared__ int total;
  if (t == 127) total = below + (flag[t] ? 1 : 0);
  __syncthreads();
  const int cnt = total;
  int32_t *ro = rows + (int64_t)item * 128;
  if (flag[t]) ro[below] = t;
  __syncthreads();
  if (!bad && cnt > 0 && cnt <= max_rows) {
    const int last = ro[cnt - 1];
    if (t >= cnt) ro[t] = last;
    for (int i = t; i < m; i += 128) cols[(int64_t)item * m + i] = rank[ix[i]];
    if (t == 0) nu[item] = cnt;
  } else {
    ro[t] = 0;
    for (int i = t; i < m; i += 128) cols[(int64_t)item * m + i] = 0;
    if (t == 0) nu[item] = -1;
  }
}

// L[item][j][i] = sum over chunks of part[chunk][item][j][cols[item][i]] for i < m, of the columns nr - 16 + (i - m) of
// the item's own rows for m <= i < m + mt   (NaN for a refused item)
__global__ __launch_bounds__(256) void latent_index_sum_kernel(const double *part, int64_t items, int k, int nr,
                                                               int nchunk, const int32_t *cols, const int32_t *nu,
                                                               int m, int mt, double *L) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int mm = m + mt;
  if (e >= items * k * mm) return;
  const int i = (int)(e % mm);
  const int64_t bj = e / mm;
  const int64_t item = bj / k;
  if (nu[item] < 0) {
    L[e] = __builtin_nan("");
    return;
  }
  const int64_t E = items * k * nr;
  const double *src = part + bj * nr + (i < m ? cols[item * m + i] : nr - 16 + (i - m));
  double a0 = 0.0, a1 = 0.0;
  int c = 0;
  for (; c + 1 < nchunk; c += 2) {
    a0 += src[(int64_t)c * E];
    a1 += src[(int64_t)(c + 1) * E];
  }
  if (c < nchunk) a0 += src[(int64_t)c * E];
  L[e] = a0 + a1;
}

template <int MC, int WV>
__global__ __launch_bounds__(WV * 64, 2) void latent_xt_kernel(LatentXtArgs A) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  constexpr int NS = LV_T / 4;                    // k-steps per tile
  constexpr int RPP = WV * 4;                     // VS^T rows staged per pass (sixteen threads per row)
  constexpr int NV = (MC * 16 + RPP - 1) / RPP;
  extern __shared__ __attribute__((aligned(16))) double smem[];   // two VS^T tiles [MC*16][XT_VLD]
  const bool want_nsq = A.nsq_part != nullptr;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15;
  const int g = lane >> 4;
  const int item = blockIdx.x;
  const int chunk = blockIdx.y;
  const int64_t nvt = (A.p + LV_T - 1) / LV_T;
  const int64_t t_lo = (int64_t)chunk * A.tiles_per_chunk;
  const int64_t t_hi = min(nvt, t_lo + A.tiles_per_chunk);
  const int srow = tid >> 4, svox = (tid & 15) * 2;
  constexpr int VT = MC * 16 * XT_VLD;             // doubles per VS^T tile

  f64x4 acc[MC];
  double nsq[MC];
#pragma unroll
  for (int mc = 0; mc < MC; ++mc) {
    acc[mc] = (f64x4){0.0, 0.0, 0.0, 0.0};
    nsq[mc] = 0.0;
  }

  uint32_t offv[NV];
#pragma unroll
  for (int q = 0; q < NV; ++q)
    offv[q] = (uint32_t)((((int64_t)min(q * RPP + srow, min(MC * 16, (int)A.k) - 1)) * A.ldv + svox) * 8);
  const char *vbase = (const char *)(A.vst + (int64_t)item * A.k * A.ldv);
  // this lane's element of a B fragment: voxel 4 s + g of the tile, data row 16 wave + col
  const double *xb = A.XT + (int64_t)g * XT_LD + 16 * wave + col;

  d2 pv[NV];
  auto fetch_v = [&](int64_t vt) {
    const int64_t vtc = min(vt, nvt - 1);                        // (a prefetch past the end re-reads the last tile)
    if ((vtc + 1) * LV_T <= A.p) {
      const char *bv = vbase + vtc * (LV_T * 8);
#pragma unroll
      for (int q = 0; q < NV; ++q) pv[q] = *(const d2 *)(bv + offv[q]);
    } else {
      const int64_t v0 = vtc * LV_T + svox;
      const int64_t c0 = (min(v0, A.p - 1) - svox) * 8, c1 = (min(v0 + 1, A.p - 1) - svox) * 8;
#pragma unroll
      for (int q = 0; q < NV; ++q)
        pv[q] = (d2){*(const double *)(vbase + offv[q] + c0), *(const double *)(vbase + offv[q] + c1)};
    }
  };
  auto park_v = [&](double *Vs) {
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      const int rr = q * RPP + srow;
      if (rr < MC * 16) {
        Vs[rr * XT_VLD + svox] = pv[q].x;
        Vs[rr * XT_VLD + svox + 1] = pv[q].y;
      }
    }
  };
  // B fragments of a tile: slot s is refilled with the NEXT tile's k-step s right after its use
  // (a whole tile of prefetch distance in eight registers)
  double bx[NS];

  if (t_lo < t_hi) {
    fetch_v(t_lo);
    {
      const double *src = xb + t_lo * (LV_T * XT_LD);
#pragma unroll
      for (int s = 0; s < NS; ++s) bx[s] = src[(int64_t)s * 4 * XT_LD];
    }
    park_v(smem);
    __syncthreads();
    int par = 0;
    for (int64_t vt = t_lo; vt < t_hi; ++vt) {
      const double *Vs = smem + par * VT;
      const double *xn = xb + min(vt + 1, nvt - 1) * (LV_T * XT_LD);   // (past the end: re-reads the last tile)
      const bool edge = (vt + 1) * LV_T > A.p;                          // (uniform) the tile holds voxels past p
      fetch_v(vt + 1);
      double a0[MC], a1[MC];
      auto lda = [&](int s, double (&a)[MC]) {
#pragma unroll
        for (int mc = 0; mc < MC; ++mc) a[mc] = Vs[(mc * 16 + col) * XT_VLD + 4 * s + g];
      };
      auto mm = [&](int s, double (&a)[MC]) {
#pragma unroll
        for (int mc = 0; mc < MC; ++mc) acc[mc] = mfma_f64(a[mc], bx[s], acc[mc]);
        bx[s] = xn[(int64_t)s * 4 * XT_LD];
      };
      // column norms: k-step s is wave (s mod WV)'s share, read once more from LDS in front of the MFMAs
      // (taken from the MFMA operands under a per-step test of the wave index, it put a branch, an fp64
      // FMA and two selects beside EVERY MFMA: 3.2 VALU instructions per MFMA in the counters);
      // voxels past p are clamped copies and do not count
      if (want_nsq) {
        for (int s = wave; s < NS; s += WV) {
          const bool in = !edge || vt * LV_T + 4 * s + g < A.p;
#pragma unroll
          for (int mc = 0; mc < MC; ++mc) {
            const double a = Vs[(mc * 16 + col) * XT_VLD + 4 * s + g];
            nsq[mc] = fma(in ? a : 0.0, a, nsq[mc]);
          }
        }
      }
      lda(0, a0);
#pragma unroll
      for (int s = 0; s < NS; s += 2) {
        lda(s + 1, a1);
        __builtin_amdgcn_sched_barrier(0);
        mm(s, a0);
        if (s + 2 < NS) lda(s + 2, a0);
        __builtin_amdgcn_sched_barrier(0);
        mm(s + 1, a1);
      }
      park_v(smem + (par ^ 1) * VT);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      par ^= 1;
    }
  }

  // Zt[j = 16 mc + g + 4 r][i = 16 wave + col]
  double *zo = A.Zt_part + ((int64_t)chunk * A.items + item) * A.k * A.n;
#pragma unroll
  for (int mc = 0; mc < MC; ++mc)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = mc * 16 + g + 4 * r;
      const int i = wave * 16 + col;
      if (j < A.k && i < A.n) zo[(int64_t)j * A.n + i] = acc[mc][r];
    }
  if (want_nsq) {
    // the waves' shares of the column norms, summed through LDS
    __syncthreads();
    double *red = smem;                            // [WV][MC*16]
#pragma unroll
    for (int mc = 0; mc < MC; ++mc) {
      double x = nsq[mc];
      x += __shfl_xor(x, 16);
      x += __shfl_xor(x, 32);
      if (g == 0) red[wave * (MC * 16) + mc * 16 + col] = x;
    }
    __syncthreads();
    if (tid < MC * 16 && tid < A.k) {
      double x = 0.0;
      for (int w = 0; w < WV; ++w) x += red[w * (MC * 16) + tid];
      A.nsq_part[((int64_t)chunk * A.items + item) * A.k + tid] = x;
    }
  }
}


// ---------------------------------------------------------------------------
// K5i: the latent scores of a SAMPLE, L_b = (X[idx_b] VS_b^T)^T, on the different rows of the sample only.
//
// A bootstrap sample of n rows holds about 0.63 n different rows, and `_compute_X_latents(X_new, V_hat)` needs no
// others (76 of 120 at config 3: five 16-row tiles instead of eight).  One WAVE = (item, range of voxels): it holds
// the accumulators of ALL the item's live row tiles x ALL tiles of latent variables (NT x MC x 4 registers) and
// streams both operands straight from global memory, each byte read once by one wave -- no LDS, no barrier:
//   * B: lane (row, g) reads voxels 16 j + 4 g .. + 3 of entry `row` of the item's sorted row list
//     (latent_index_meta_kernel) from a TILE-MAJOR copy of X, XB[tile of 32 voxels][row][32] (voxels past p zero;
//     xb_prepare_kernel, once per X): one 32-byte load per lane = one whole 128-byte line per row and instruction,
//     the row choice is the lane's own offset -- no gather;
//   * A: lane (latent variable, g) reads the same four voxels of its row of VS^T; k-step e of the group multiplies
//     voxel 16 j + 4 g + e on both sides.
// A group = MC + NT loads of 32 bytes per lane, 4 NT MC MFMAs.  The VS^T operands (from HBM) sit in two register sets
// used in turn, two groups of prefetch; a row tile's B registers (XB: from L2) are refilled right after its MFMAs.  The item's number of live tiles is a wave-uniform switch
// over instances of the loop (NT = 1 .. 8).  Workgroup id -> (item, range) keeps the ranges of an XCD apart from the
// other XCDs' (id mod 8 = range mod 8) and its resident waves on the same ranges of different items: XB streams
// through that XCD's L2 once per batch of items.
//
// What came before (config 3, us per item; K5x on all 128 rows: 39-42): K5x's workgroup (one wave per row tile, VS^T
// through LDS) on the row lists -- through X^T 43, through X itself 54, through XB 42, with two tiles of prefetch
// 40.  Ablation builds (`microbench/latent_index_once.py`): without the MFMAs 27, and without any one of B loads /
// VS^T loads / LDS reads still 43 -- the six waves of a workgroup (five live) sit 2-2-1-1 on the four SIMDs, the
// second workgroup of the CU the same way, so one SIMD carries four live waves where 2.5 would be its share.  A
// wave per (item, voxel range) is balanced whatever the number of live tiles.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void xb_prepare_kernel(const double *X, int64_t ldx, int64_t p, int n, double *XB) {
  // one thread per pair of voxels: XB[t][r][2 q .. 2 q + 1], 16 threads per row of a tile
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t nvt = (p + LV_T - 1) / LV_T;
  if (e >= nvt * n * 16) return;
  const int q = (int)(e & 15);
  const int64_t tr = e >> 4;
  const int r = (int)(tr % n);
  const int64_t v = (tr / n) * LV_T + 2 * q;
  const double *src = X + (int64_t)r * ldx;
  XB[2 * e] = v < p ? src[v] : 0.0;
  XB[2 * e + 1] = v + 1 < p ? src[v + 1] : 0.0;
}

struct LatentWaveArgs {
  const double *XB;       // [tiles][n][32]
  int64_t p;
  int32_t n, k, items;
  const double *vst;      // [items][k][ldv]; tiled: [items][ldv / 32][k][32]
  int64_t ldv;
  int32_t vst_tiled;
  int32_t nsplit;         // voxel ranges per item (a multiple of 8)
  int32_t tiles_per_split;
  double *Zt_part;        // [nsplit][items][k][nr]
  double *nsq_part;       // [nsplit][items][k] or null
  const int32_t *rows;    // [items][XT_LD]
  const int32_t *nu;      // [items]
  int32_t nr;             // columns of Zt_part (a multiple of 16; with T: the last sixteen are its columns)
  int32_t tile_lo;        // first row tile of this launch (see latent_wave_kernel)
  // optional extra tile of B rows that are the item's OWN rows (the multiblock's raw task rows): row t of the tile is
  // row t_row[t] of the item's block of T, row-major; its products land in the last sixteen columns
  const double *T;        // [items][t_item_rows][t_ld] or null
  int64_t t_ld;
  int32_t t_item_rows, t_rows;
  const int32_t *t_row;   // [t_rows] (device)
};

// NT tiles of B rows: NT - TT from the item's row list (entries 16 (tile_lo + i) ..), and with TT the tile of the item's
// own rows last.
template <int MC, int NT, bool TT>
__device__ __forceinline__ void latent_wave_body(const LatentWaveArgs &A, int item, int split, int lane) {
  constexpr int NX = NT - (TT ? 1 : 0);
  typedef double d2 __attribute__((ext_vector_type(2)));
  const int col = lane & 15;
  const int g = lane >> 4;
  const int64_t nvt = (A.p + LV_T - 1) / LV_T;
  const int64_t t_lo = (int64_t)split * A.tiles_per_split;
  const int64_t t_hi = min(nvt, t_lo + A.tiles_per_split);

  f64x4 acc[NT][MC];
  double nsq[MC];
#pragma unroll
  for (int mc = 0; mc < MC; ++mc) {
    nsq[mc] = 0.0;
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[i][mc] = (f64x4){0.0, 0.0, 0.0, 0.0};
  }
  // 32-bit byte offsets of this lane's rows from the (scalar) start of a tile (the library checks the sizes)
  uint32_t offa[MC], offb[NT];
#pragma unroll
  for (int mc = 0; mc < MC; ++mc) offa[mc] = (uint32_t)(((int64_t)min(mc * 16 + col, A.k - 1) * (A.vst_tiled ? LV_T : A.ldv) + 2 * g) * 8);
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int r = min(max(A.rows[(int64_t)item * XT_LD + 16 * (A.tile_lo + i) + col], 0), A.n - 1);
    offb[i] = (uint32_t)((r * LV_T + 2 * g) * 8);
  }
  if (TT) offb[NT - 1] = (uint32_t)(((int64_t)A.t_row[min(col, A.t_rows - 1)] * A.t_ld + 2 * g) * 8);
  // Both operands through buffer descriptors based at the range's first tile: a load is descriptor + this lane's 32-bit
  // row offset + a scalar tile offset (+ immediate), no 64-bit address arithmetic in registers (with plain pointers
  // the compiler kept a 64-bit address pair per row tile and doubled the B registers: 250 VGPRs and scratch for what
  // needs 110).  (The record counts are unsigned 32-bit: an item's VS^T may span up to 4 GiB.)
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const char *vbase = (const char *)(A.vst + (int64_t)item * A.k * A.ldv);
  const int64_t xts = (int64_t)A.n * LV_T * 8;                    // bytes per tile of XB
  // bytes from one tile of VS^T to the next: row-major 256 (a row's next 32 voxels), tile-major the tile's k rows
  const int64_t vts = A.vst_tiled ? (int64_t)A.k * LV_T * 8 : LV_T * 8;
  const int64_t vrange = (int64_t)A.k * A.ldv * 8 - t_lo * vts;
  const int64_t xrange = (nvt - t_lo) * xts;
  const auto vsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(vbase + t_lo * vts), 0,
                                                      (int)(uint32_t)min(vrange, (int64_t)0xffffffff), 0x00020000);
  const auto xsrc = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)A.XB + t_lo * xts), 0,
                                                      (int)(uint32_t)min(xrange, (int64_t)0xffffffff), 0x00020000);
  const char *tbase = TT ? (const char *)(A.T + (int64_t)item * A.t_item_rows * A.t_ld) : nullptr;
  const int64_t trange = TT ? (int64_t)A.t_item_rows * A.t_ld * 8 - t_lo * (LV_T * 8) : 0;
  const auto tsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(tbase + (TT ? t_lo * (LV_T * 8) : 0)), 0,
                                                      (int)(uint32_t)min(trange, (int64_t)0xffffffff), 0x00020000);
  auto load16 = [&](decltype(vsrc) src, uint32_t voff, uint32_t soff) -> d2 {
    const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(src, voff, soff, 0);
    union {
      u32x4 u;
      d2 d;
    } cv;
    cv.u = u;
    return cv.d;
  };

  // group (tile t, j = 0 .. 3): voxels 8 j .. 8 j + 7 of the tile, two k-steps: lane (row, g) holds voxels 8 j + 2 g + {0, 1}
  // of its row (16 bytes; the four lanes of a row read 64 contiguous bytes).  The loop runs over the WHOLE tiles of the
  // range (its prefetches stop at the last of them); the tile that holds voxels past p, if the range ends with it,
  // follows with element-wise clamped loads of VS^T (its voxels past p are zero in XB).
  const int64_t t_full = min(t_hi, A.p / LV_T);
  auto mfmas = [&](const d2 (&a)[MC], const d2 &b, int i) {
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int mc = 0; mc < MC; ++mc) acc[i][mc] = mfma_f64(a[mc][e], b[e], acc[i][mc]);
  };
  if (t_lo < t_full) {
    // Two groups per loop iteration, the two A sets (group in use / two groups on) with static names.
    d2 a0[MC], a1[MC], b[NT];
#pragma unroll
    for (int mc = 0; mc < MC; ++mc) a0[mc] = load16(vsrc, offa[mc], 0);
#pragma unroll
    for (int i = 0; i < NT; ++i) b[i] = load16(TT && i == NT - 1 ? tsrc : xsrc, offb[i], 0);
#pragma unroll
    for (int mc = 0; mc < MC; ++mc) a1[mc] = load16(vsrc, offa[mc], 64);
    const int nq = 4 * (int)(t_full - t_lo);                     // (even)
    const uint32_t xts32 = (uint32_t)xts, vts32 = (uint32_t)vts;
    // one group: MFMAs row tile by row tile, the tile's B registers refilled (next group: XB comes from L2) right behind
    // them; the column norms of VS^T and the prefetch of group q + 2's A operands (VS^T comes from HBM) at the end; the
    // last groups' prefetches re-read the last group
    auto group = [&](int q, d2 (&a)[MC]) {
      const int q1 = min(q + 1, nq - 1), q2 = min(q + 2, nq - 1);
      const uint32_t xnext = (uint32_t)(q1 >> 2) * xts32 + (uint32_t)(q1 & 3) * 64;
      const uint32_t vnext = (uint32_t)(q2 >> 2) * vts32 + (uint32_t)(q2 & 3) * 64;
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        mfmas(a, b[i], i);
        b[i] = TT && i == NT - 1 ? load16(tsrc, offb[i], (uint32_t)q1 * 64) : load16(xsrc, offb[i], xnext);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int mc = 0; mc < MC; ++mc) {
        nsq[mc] = fma(a[mc][0], a[mc][0], nsq[mc]);
        nsq[mc] = fma(a[mc][1], a[mc][1], nsq[mc]);
        a[mc] = load16(vsrc, offa[mc], vnext);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    // the prologue's loads are drained here: with them pending, the wait-count pass merges their state into the loop
    // header and every iteration starts with vmcnt(1), i.e. behind the A loads it has just issued
    __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma nounroll
    for (int q = 0; q < nq; q += 2) {
      group(q, a0);
      group(q + 1, a1);
    }
  }
  if (t_lo < t_hi && t_full < t_hi) {
    const int64_t t = t_full;                                     // == nvt - 1, p % 32 != 0
    for (int j = 0; j < 4; ++j) {
      d2 a[MC];
#pragma unroll
      for (int mc = 0; mc < MC; ++mc) {
        // (row-major: the row's voxel v at 8 v; tile-major: this tile's row at t vts, voxel v mod 32)
        const char *rowp = vbase + offa[mc] - 2 * g * 8 + (A.vst_tiled ? t * vts - t * (LV_T * 8) : 0);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int64_t v = t * LV_T + 8 * j + 2 * g + e;
          const double x = *(const double *)(rowp + min(v, A.p - 1) * 8);
          a[mc][e] = v < A.p ? x : 0.0;
          nsq[mc] = fma(a[mc][e], a[mc][e], nsq[mc]);
        }
      }
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        const d2 b = *(const d2 *)((const char *)A.XB + t * xts + offb[i] + j * 64);
        mfmas(a, b, i);
      }
      if (TT) {
        const char *rowp = tbase + offb[NT - 1] - 2 * g * 8;
        d2 b;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int64_t v = t * LV_T + 8 * j + 2 * g + e;
          const double x = *(const double *)(rowp + min(v, A.p - 1) * 8);
          b[e] = v < A.p ? x : 0.0;
        }
        mfmas(a, b, NT - 1);
      }
    }
  }

  // Zt[j = 16 mc + g + 4 r][i' = 16 i + col]
  double *zo = A.Zt_part + ((int64_t)split * A.items + item) * A.k * A.nr;
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int mc = 0; mc < MC; ++mc)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = mc * 16 + g + 4 * r;
        if (j < A.k) zo[(int64_t)j * A.nr + (TT && i == NT - 1 ? A.nr - 16 : 16 * (A.tile_lo + i)) + col] = acc[i][mc][r];
      }
  if (A.nsq_part) {
#pragma unroll
    for (int mc = 0; mc < MC; ++mc) {
      double x = nsq[mc];
      x += __shfl_xor(x, 16);
      x += __shfl_xor(x, 32);
      const int j = mc * 16 + col;
      if (g == 0 && j < A.k) A.nsq_part[((int64_t)split * A.items + item) * A.k + j] = x;
    }
  }
}

// A wave holds at most latent_wave_cap(MC) row tiles (NT x MC x 8 accumulator registers of the 256 at two waves per
// SIMD).  Items with more live tiles (a wide sample; k > 48) get the rest from a second launch with tile_lo = cap,
// whose waves for all other items leave at once.
constexpr int latent_wave_cap(int mc) { return mc == 1 ? 8 : mc == 2 ? 7 : 18 / mc; }   // (MC = 2 with eight tiles: 8 bytes of scratch)

template <int MC>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void latent_wave_kernel(LatentWaveArgs A) {
  constexpr int CAP = latent_wave_cap(MC);
  const int lane = threadIdx.x;
  // id mod 8 (the XCD) = split mod 8; consecutive ids of an XCD: the items of one split
  const int64_t id = blockIdx.x;
  const int x = (int)(id & 7);
  const int64_t q = id >> 3;
  const int item = (int)(q % A.items);
  const int split = (int)(q / A.items) * 8 + x;
  const int nu = A.nu[item];
  // the item's tiles: its row tiles, then (with T) the tile of its own rows; this launch takes tile_lo .. tile_lo + CAP
  const int nx = nu < 0 ? -1 : (nu + 15) >> 4;
  const bool with_t = __builtin_amdgcn_readfirstlane(A.T != nullptr && nx >= A.tile_lo && nx < A.tile_lo + CAP ? 1 : 0) != 0;
  const int nt = __builtin_amdgcn_readfirstlane(nx < 0 ? 0 : min(nx + (A.T != nullptr ? 1 : 0) - A.tile_lo, CAP));
#define PLSR_WB(N)                                                    \
  case N:                                                             \
    if constexpr (CAP >= N) {                                         \
      if (with_t)                                                     \
        latent_wave_body<MC, N, true>(A, item, split, lane);          \
      else                                                            \
        latent_wave_body<MC, N, false>(A, item, split, lane);         \
    }                                                                 \
    break;
  switch (nt) {
    PLSR_WB(1) PLSR_WB(2) PLSR_WB(3) PLSR_WB(4) PLSR_WB(5) PLSR_WB(6) PLSR_WB(7) PLSR_WB(8)
    default: break;                                  // (nothing for this launch, or a refused item: NaN from the sum kernel)
  }
#undef PLSR_WB
}

}  // namespace plsr
