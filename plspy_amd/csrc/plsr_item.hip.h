// K4 / K5: bootstrap of behaviour and multiblock PLS, where every resample has
// its OWN data matrix (rows of X gathered and z-scored within the resample's
// cells by K3), so resamples cannot share one X tile as in K1.
//
// K4  item_project_kernel<MC>   (pass A, voxel-major)
//     VS_b[j, v] = sum_i op_b[j, i] * Z_b[i, v]              (k x p per item)
//     One workgroup owns 64 voxels for ALL items of the launch, so the
//     bootstrap moments  S1 += VS - ref,  S2 += (VS - ref)^2  stay in
//     registers across items (bootstrap_permutation.py:620-626, :695 without
//     the R x p x k stack).  VS_b^T is also written out ([item][j][v]) for K5.
//     MFMA: M = 16 latent variables, N = 16 voxels (one block per wave), K = 4
//     rows; operator fragments and the item's X rows are staged per item in
//     K-chunks through LDS.
//
// K5  latent_kernel<MC>         (pass B, item-major)
//     Zt_b[j, i] = sum_v VS_b[j, v] * X[i, v]                (k x n per item)
//     nsq_b[j]   = sum_v VS_b[j, v]^2
//     i.e. X @ VS_b (class_functions.py:165-182 as used at
//     bootstrap_permutation.py:638, :647, :655 before the column
//     normalisation) and the norms of :623.  One workgroup owns (item, voxel
//     chunk) and accumulates over its voxels in registers; chunk partials are
//     summed afterwards in fixed order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "plsr_project.hip.h"

namespace plsr {

struct ItemArgs {
  const double *Z;        // [items][n][ldz]  per-item matrices
  int64_t z_item_stride, ldz, p;
  int32_t n, nk, ks;      // rows, k-steps, k-steps staged per chunk
  const double *frag;     // [items*MC][nk][64] rows layout (ops_rows_kernel)
  int32_t items, k;
  const double *ref;      // [p][k] shift of the moment sums, or null
  double *S1, *S2;        // [split][p][k] partial sums (overwritten)
  double *vst;            // [items][k][ldv]  VS^T, or null
  int64_t ldv;
};

template <int MC>
__global__ __launch_bounds__(256, 2) void item_project_kernel(ItemArgs A) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15;
  const int g = lane >> 4;
  const int64_t v0 = (int64_t)blockIdx.x * TV;
  const int64_t v = v0 + wave * 16 + col;                // this lane's voxel
  double *ops = smem;                                     // [MC][ks][64]
  double *Xs = smem + (size_t)MC * A.ks * 64;             // [4*ks][64]
  const int xo = xs_index(g, wave * 16 + col);

  // items are split over blockIdx.y; each split owns its moment partials
  const int per = (A.items + gridDim.y - 1) / gridDim.y;
  const int it_lo = blockIdx.y * per;
  const int it_hi = min(A.items, it_lo + per);

  double s1[MC][4], s2[MC][4], rf[MC][4];
#pragma unroll
  for (int mc = 0; mc < MC; ++mc)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = mc * 16 + g + 4 * r;
      s1[mc][r] = 0.0;
      s2[mc][r] = 0.0;
      rf[mc][r] = (A.ref != nullptr && j < A.k && v < A.p) ? A.ref[v * A.k + j] : 0.0;
    }

  for (int item = it_lo; item < it_hi; ++item) {
    const double *Zi = A.Z + (int64_t)item * A.z_item_stride;
    const double *fi = A.frag + (int64_t)item * MC * A.nk * 64;
    f64x4 D[MC];
#pragma unroll
    for (int mc = 0; mc < MC; ++mc) D[mc] = (f64x4){0.0, 0.0, 0.0, 0.0};
    for (int ks0 = 0; ks0 < A.nk; ks0 += A.ks) {
      const int ks1 = min(A.nk, ks0 + A.ks);
      __syncthreads();
      for (int r0 = 4 * ks0; r0 < 4 * ks1; r0 += 16) {
        double tmp[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int row = r0 + u * 4 + (tid >> 6);
          const int64_t vv = v0 + lane;
          tmp[u] = (row < A.n && vv < A.p) ? Zi[(int64_t)row * A.ldz + vv] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int row = r0 + u * 4 + (tid >> 6);
          if (row < 4 * ks1) Xs[xs_index(row - 4 * ks0, lane)] = tmp[u];
        }
      }
      for (int e = tid; e < MC * (ks1 - ks0) * 64; e += 256) {
        const int mc = e / ((ks1 - ks0) * 64);
        const int rest = e % ((ks1 - ks0) * 64);
        ops[((size_t)mc * A.ks) * 64 + rest] = fi[((size_t)mc * A.nk + ks0) * 64 + rest];
      }
      __syncthreads();
      for (int s = 0; s < ks1 - ks0; ++s) {
        const double b = Xs[(size_t)s * 4 * TV + xo];
#pragma unroll
        for (int mc = 0; mc < MC; ++mc) D[mc] = mfma_f64(ops[((size_t)mc * A.ks + s) * 64 + lane], b, D[mc]);
      }
    }
    // D[mc][r] = VS[j = 16 mc + g + 4 r][v]
#pragma unroll
    for (int mc = 0; mc < MC; ++mc)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = mc * 16 + g + 4 * r;
        const double d = D[mc][r] - rf[mc][r];
        s1[mc][r] += d;
        s2[mc][r] = fma(d, d, s2[mc][r]);
        if (A.vst != nullptr && j < A.k && v < A.p) A.vst[((int64_t)item * A.k + j) * A.ldv + v] = D[mc][r];
      }
  }

  if (v < A.p) {
    double *o1 = A.S1 + (int64_t)blockIdx.y * A.p * A.k;
    double *o2 = A.S2 + (int64_t)blockIdx.y * A.p * A.k;
#pragma unroll
    for (int mc = 0; mc < MC; ++mc)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = mc * 16 + g + 4 * r;
        if (j < A.k) {
          o1[v * A.k + j] = s1[mc][r];
          o2[v * A.k + j] = s2[mc][r];
        }
      }
  }
}

inline size_t item_lds_bytes(int mc, int ks) { return ((size_t)mc * ks * 64 + (size_t)ks * 4 * TV) * sizeof(double); }

// ---------------------------------------------------------------------------
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
constexpr int LV_LD = 34;     // padded LDS row (conflict-free row-strided ds_read_b64)

// MC = 16-row tiles of latent variables; NI = 16-row tiles of data rows per wave;
// IG = items per workgroup: they share every staged X tile (X is re-read once
// per IG items instead of once per item) and the two barriers per tile;
// WV = waves per workgroup (each owns NI tiles of data rows).
// The global loads of tile t+1 are issued before the MFMAs of tile t and parked
// in registers, so only the LDS write sits between the two barriers.
template <int MC, int NI, int IG, int WV>
__global__ __launch_bounds__(WV * 64, 2) void latent_kernel(LatentArgs A) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15;
  const int g = lane >> 4;
  const int item0 = blockIdx.x * IG;
  const int chunk = blockIdx.y;
  constexpr int nrow_x = WV * NI * 16;            // every wave's NI tiles of data rows (zero padded)
  constexpr int RPP = WV * 2;                     // rows staged per pass (32 voxels per row)
  constexpr int NV = (IG * MC * 16 + RPP - 1) / RPP;   // passes over the VS^T rows
  constexpr int NX = nrow_x / RPP;                     // passes over the X rows
  double *Vs = smem;                              // [IG][MC*16][LV_LD]   VS^T tiles
  double *Xs = smem + (size_t)IG * MC * 16 * LV_LD;    // [nrow_x][LV_LD]  X tile
  const int64_t nvt = (A.p + LV_T - 1) / LV_T;
  const int64_t t_lo = (int64_t)chunk * A.tiles_per_chunk;
  const int64_t t_hi = min(nvt, t_lo + A.tiles_per_chunk);
  const int srow = tid >> 5, svox = tid & 31;

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

  double pv[NV], px[NX];
  auto fetch = [&](int64_t vt) {
    const int64_t vv = vt * LV_T + svox;
    const bool inv = vv < A.p && vt < t_hi;
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      const int rr = q * RPP + srow;                        // row of the stacked [IG][MC*16] tile
      const int ig = rr / (MC * 16), row = rr % (MC * 16);
      const int item = min(item0 + ig, A.items - 1);        // a short last group recomputes the last item
      pv[q] = (inv && rr < IG * MC * 16 && row < A.k) ? A.vst[((int64_t)item * A.k + row) * A.ldv + vv] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < NX; ++q) {
      const int row = q * RPP + srow;
      px[q] = (inv && row < A.n) ? A.X[(int64_t)row * A.ldx + vv] : 0.0;
    }
  };
  auto park = [&]() {
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      const int rr = q * RPP + srow;
      if (rr < IG * MC * 16) Vs[rr * LV_LD + svox] = pv[q];
    }
#pragma unroll
    for (int q = 0; q < NX; ++q) Xs[(q * RPP + srow) * LV_LD + svox] = px[q];
  };

  fetch(t_lo);
  for (int64_t vt = t_lo; vt < t_hi; ++vt) {
    __syncthreads();                  // everybody is done reading the previous tile
    park();
    __syncthreads();
    fetch(vt + 1);                    // in flight during the MFMAs below
#pragma unroll
    for (int s = 0; s < LV_T / 4; ++s) {
      double b[NI];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int it = wave * NI + ni;                          // this wave's tiles of data rows
        b[ni] = Xs[(it * 16 + col) * LV_LD + 4 * s + g];            // B[k = v][n = i]
      }
#pragma unroll
      for (int ig = 0; ig < IG; ++ig) {
#pragma unroll
        for (int mc = 0; mc < MC; ++mc) {
          const double a = Vs[((ig * MC + mc) * 16 + col) * LV_LD + 4 * s + g];   // A[m = j][k = v]
          nsq[ig][mc] = fma(a, a, nsq[ig][mc]);
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) acc[ig][mc][ni] = mfma_f64(a, b[ni], acc[ig][mc][ni]);
        }
      }
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
    if (wave == 0) {
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

}  // namespace plsr
