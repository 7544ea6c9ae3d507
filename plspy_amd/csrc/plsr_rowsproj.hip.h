// K4m: the multiblock bootstrap's projection from the un-normalised rows' products (round 3).
//
// A multiblock bootstrap normalises the rows of its cross-block over ALL voxels before projecting them
// (class_functions.py:503-505, bootstrap_permutation.py:610, :620):  VS_b = (D_b^-1 raw_b Z_b)^T U.  The norms
// D_b are known only after a pass over all voxels, so round 2 ran K4a twice per batch -- a norms-only pass with
// the raw rows as operator, then the projection with the operator U^T D_b^-1 raw_b (3 GFLOP per item each).
// Here the first pass also stores its products R_b = raw_b Z_b (kr x p, 61 MB per item at config 6) and the
// second pass is a stream over them:
//
//     VS_b[j, v] = sum_r U[r, j] / |row_r|_b * R_b[r, v]          0.6 GFLOP per item, 2 x 61 MB of HBM traffic
//
// in place (R_b's rows become VS_b^T for the latent kernel), with K4a's moment epilogue (plain sums of VS and
// VS^2 per voxel over the items of a split; the shift by the observed VS is applied at the merge,
// moment_unshift_kernel).  A wave owns 16 voxels for a run of items: the rows' products of item b + 1 are
// requested before the MFMAs of item b, U^T lives in registers as A fragments, the inverse norms come from a
// small table.  Bound: HBM (122 MB per item: 24 us at the 5 TB/s of a copy).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "plsr_project.hip.h"

namespace plsr {

struct RowsProjArgs {
  double *R;                 // [items][kr][ldv]  in: R_b, out (when `out` is null): rows 0 .. k - 1 = VS_b^T
  double *out;               // [items][k][ldv] VS_b^T, or null: in place
  int64_t ldv, p;
  int32_t items, kr, k, ks, per;   // ks = ceil(kr / 4) k-steps; per = items per split
  const double *rdinv;       // [items][4 ks]  1 / |row_r|_b (0 for a row of norm 0 and past kr)
  const double *ufrag;       // [MC][ks][64]   A fragments of U^T: lane (j, kk) = U[4 s + kk][16 mc + j]
  double *S1, *S2;           // [split][p][k] plain partial sums (overwritten) or null
};

__global__ __launch_bounds__(256) void rows_project_meta_kernel(const double *rowsq, int64_t rowsq_stride, const double *U,
                                                                int items, int kr, int k, int ks, int MC, double *rdinv,
                                                                double *ufrag) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t n1 = (int64_t)items * 4 * ks, n2 = (int64_t)MC * ks * 64;
  if (e < n1) {
    const int r = (int)(e % (4 * ks));
    const int64_t b = e / (4 * ks);
    const double q = r < kr ? rowsq[b * rowsq_stride + r] : 0.0;
    rdinv[e] = q > 0.0 ? 1.0 / sqrt(q) : 0.0;
  } else if (e - n1 < n2) {
    const int64_t f = e - n1;
    const int lane = (int)(f & 63), s = (int)((f >> 6) % ks), mc = (int)((f >> 6) / ks);
    const int j = 16 * mc + (lane & 15), r = 4 * s + (lane >> 4);
    ufrag[f] = (j < k && r < kr) ? U[(int64_t)r * k + j] : 0.0;
  }
}

template <int MC, int KSMAX>
__global__ __launch_bounds__(256, 2) void rows_project_kernel(RowsProjArgs A) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int col = lane & 15, g = lane >> 4;
  const int64_t v = ((int64_t)blockIdx.x * 4 + wave) * 16 + col;
  const bool vok = v < A.p;
  const int64_t vc = vok ? v : A.p - 1;
  const int it_lo = blockIdx.y * A.per, it_hi = min(A.items, it_lo + A.per);
  const int ks = A.ks;

  // U^T as A fragments in LDS, shared by the workgroup's four waves (in registers, beside the moment sums and the
  // two sets of rows, the k = 38 instance spilled 132 bytes per lane at two waves per SIMD); this lane's row
  // offsets (bytes, < 4 GiB: the library checks) for loads and stores
  __shared__ double ufs[MC * KSMAX * 64];
  for (int e = threadIdx.x; e < MC * KSMAX * 64; e += 256) {
    const int s = (e >> 6) % KSMAX, mc = (e >> 6) / KSMAX;
    ufs[e] = s < ks ? A.ufrag[((size_t)mc * ks + s) * 64 + (e & 63)] : 0.0;
  }
  __syncthreads();
  const double *uf = ufs + lane;
  uint32_t lo[KSMAX], so[MC][4];
#pragma unroll
  for (int s = 0; s < KSMAX; ++s) lo[s] = (uint32_t)(((int64_t)min(4 * s + g, A.kr - 1) * A.ldv + vc) * 8);
#pragma unroll
  for (int mc = 0; mc < MC; ++mc)
#pragma unroll
    for (int r = 0; r < 4; ++r) so[mc][r] = (uint32_t)(((int64_t)min(16 * mc + g + 4 * r, A.kr - 1) * A.ldv + vc) * 8);

  if (it_lo >= it_hi) return;
  double s1[MC][4], s2[MC][4];
#pragma unroll
  for (int mc = 0; mc < MC; ++mc)
#pragma unroll
    for (int r = 0; r < 4; ++r) s1[mc][r] = s2[mc][r] = 0.0;

  const int64_t item_bytes = (int64_t)A.kr * A.ldv * 8;
  char *base = (char *)A.R + (int64_t)it_lo * item_bytes;
  // results: over R_b (its rows of this voxel column were all read before the first MFMA could issue), or to a
  // block of their own (R_b stays: the latent kernel reads the raw task rows from it)
  const int64_t out_bytes = A.out ? (int64_t)A.k * A.ldv * 8 : item_bytes;
  char *obase = A.out ? (char *)A.out + (int64_t)it_lo * out_bytes : base;
  const double *rdp = A.rdinv + (size_t)it_lo * 4 * ks + g;
  double rn[KSMAX], dn[KSMAX];
#pragma unroll
  for (int s = 0; s < KSMAX; ++s) {
    rn[s] = *(const double *)(base + lo[s]);
    dn[s] = rdp[s < ks ? 4 * s : 0];
  }
  for (int item = it_lo; item < it_hi; ++item) {
    double b[KSMAX];
#pragma unroll
    for (int s = 0; s < KSMAX; ++s) b[s] = rn[s] * dn[s];
    // the next item's rows (the last item re-reads itself: values unused)
    char *nb = item + 1 < it_hi ? base + item_bytes : base;
    const double *nd = item + 1 < it_hi ? rdp + 4 * ks : rdp;
#pragma unroll
    for (int s = 0; s < KSMAX; ++s) {
      rn[s] = *(const double *)(nb + lo[s]);
      dn[s] = nd[s < ks ? 4 * s : 0];
    }
    f64x4 acc[MC];
#pragma unroll
    for (int mc = 0; mc < MC; ++mc) acc[mc] = (f64x4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < KSMAX; ++s)
      if (s < ks) {
#pragma unroll
        for (int mc = 0; mc < MC; ++mc) acc[mc] = mfma_f64(uf[(mc * KSMAX + s) * 64], b[s], acc[mc]);
      }
    // acc[mc][r] = VS_b[row 16 mc + g + 4 r][voxel]: moments, then the store
#pragma unroll
    for (int mc = 0; mc < MC; ++mc)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double val = acc[mc][r];
        s1[mc][r] += val;
        s2[mc][r] = fma(val, val, s2[mc][r]);
        if (vok && 16 * mc + g + 4 * r < A.k) *(double *)(obase + so[mc][r]) = val;
      }
    base = nb;
    obase += out_bytes;
    rdp = nd;
  }
  if (A.S1 != nullptr && vok) {
    double *o1 = A.S1 + ((int64_t)blockIdx.y * A.p + v) * A.k;
    double *o2 = A.S2 + ((int64_t)blockIdx.y * A.p + v) * A.k;
#pragma unroll
    for (int mc = 0; mc < MC; ++mc)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = 16 * mc + g + 4 * r;
        if (j < A.k) {
          o1[j] = s1[mc][r];
          o2[j] = s2[mc][r];
        }
      }
  }
}

}  // namespace plsr
