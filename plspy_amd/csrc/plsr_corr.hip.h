// K3: row gather + per-cell z-scoring (the correlation preprocess of behaviour
// and multiblock PLS).
//
//   out[item][r][v] = X[src[item][r]][v]                          (raw cells)
//   out[item][r][v] = (X[src[r]][v] - mu_c[v]) / (sd_c[v] sqrt(n_c))   (z cells)
//
// with mu_c / sd_c the mean and ddof-0 standard deviation over the rows of the
// output cell c that contains r.  A voxel whose sd <= eps*|mu| in a cell gives 0
// in that cell (scipy.stats.zscore's constant-slice rule followed by the
// reference's nan_to_num).  Contracting the result with per-cell z-scored
// behaviour columns gives class_functions.py:185-247 (_compute_corr):
//   R_cell = Yz.T @ Xz.
// One thread per voxel (rows of X are read coalesced), blockIdx.y = item.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace plsr {

struct GatherArgs {
  const double *X;         // [n][ldx]
  int64_t ldx, p;
  const int32_t *src;      // [items][nout] source row of every output row
  const int32_t *cell_lo;  // [ncell+1] output-row ranges of the cells (shared by all items)
  const int32_t *cell_z;   // [ncell] 1 = z-score the cell, 0 = copy rows
  int32_t nout, ncell;
  double *out;             // [items][nout][ldo]
  int64_t ldo;
};

__global__ __launch_bounds__(256) void gather_zscore_kernel(GatherArgs A) {
  const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (v >= A.p) return;
  const int item = blockIdx.y;
  const int32_t *src = A.src + (int64_t)item * A.nout;
  double *out = A.out + (int64_t)item * A.nout * A.ldo;
  for (int c = 0; c < A.ncell; ++c) {
    const int lo = A.cell_lo[c], hi = A.cell_lo[c + 1];
    if (!A.cell_z[c]) {
      for (int r = lo; r < hi; ++r) out[(int64_t)r * A.ldo + v] = A.X[(int64_t)src[r] * A.ldx + v];
      continue;
    }
    const double cnt = (double)(hi - lo);
    double mu = 0.0;
    for (int r = lo; r < hi; ++r) mu += A.X[(int64_t)src[r] * A.ldx + v];
    mu /= cnt;
    double var = 0.0;
    for (int r = lo; r < hi; ++r) {
      const double d = A.X[(int64_t)src[r] * A.ldx + v] - mu;
      var = fma(d, d, var);
    }
    const double sd = sqrt(var / cnt);
    const bool dead = !(sd > 2.220446049250313e-16 * fabs(mu));
    const double scale = dead ? 0.0 : 1.0 / (sd * sqrt(cnt));
    for (int r = lo; r < hi; ++r)
      out[(int64_t)r * A.ldo + v] = dead ? 0.0 : (A.X[(int64_t)src[r] * A.ldx + v] - mu) * scale;
  }
}

// K0: out = rows @ X for a handful of operator rows (m x n) -- the observed blocks of a
// PLS() call: cell means and the mean-centred block (class_functions.py:7-95 as the
// operator W), the behaviour correlation block (:185-247 on the z-scored X), the
// multiblock (:454-516), contrast projections (:126-162) and the back-projection
// V = M^T U / s of the thin SVD (:98-123).  HBM-bound: one thread per voxel reads the
// n rows of X coalesced, once per slice of 16 output rows; the slice's operator
// entries sit transposed in LDS ([i][16], every lane reads the same words).  No MFMA:
// 2 m n p flops is below 0.5 GFLOP here, the 8 n p bytes of X are what costs.
__global__ __launch_bounds__(256) void rows_apply_kernel(const double *__restrict__ X, int64_t ldx, int64_t p,
                                                        int n, const double *__restrict__ rows, int m,
                                                        double *__restrict__ out, int64_t ldo) {
  extern __shared__ __attribute__((aligned(16))) double rs[];     // [n][16]
  const int m0 = blockIdx.y * 16;
  const int mc = min(16, m - m0);
  for (int e = threadIdx.x; e < n * 16; e += 256) {
    const int i = e >> 4, r = e & 15;
    rs[e] = r < mc ? rows[(int64_t)(m0 + r) * n + i] : 0.0;
  }
  __syncthreads();
  const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (v >= p) return;
  double acc[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0;
  const double *xp = X + v;
  int i = 0;
  for (; i + 4 <= n; i += 4) {
    double x[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) x[u] = xp[(int64_t)(i + u) * ldx];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = fma(rs[(i + u) * 16 + r], x[u], acc[r]);
  }
  for (; i < n; ++i) {
    const double x = xp[(int64_t)i * ldx];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = fma(rs[i * 16 + r], x, acc[r]);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r)
    if (r < mc) out[(int64_t)(m0 + r) * ldo + v] = acc[r];
}

}  // namespace plsr
