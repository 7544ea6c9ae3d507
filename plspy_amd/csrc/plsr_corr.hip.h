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

}  // namespace plsr
