// K4f: bootstrap of behaviour / multiblock PLS without materialising the
// per-resample matrix.
//
// Every resample ("item") b has its own data matrix Z_b: rows of X gathered by
// src_b and z-scored within output-row cells (class_functions.py:185-247),
// and its own operator rows op_b (k x nz):
//
//     VS_b[j, v] = sum_i op_b[j, i] * Z_b[i, v],
//     Z_b[i, v]  = X[src_b[i], v] * sc_b,c(v) + sh_b,c(v)      (i in cell c)
//
// with sc = 1 / (sd sqrt(n_c)), sh = -mean * sc for z-scored cells and
// (1, 0) for copied cells.  A workgroup owns 64 voxels for ALL items of the
// launch and keeps the whole X[:, tile] (n x 64) in LDS, so an item costs no
// HBM traffic for X at all:
//
//   item_stats_kernel   two-pass mean / sd of the gathered rows per (item,
//                       cell, voxel) from the LDS tile -> sc, sh
//   item_meta_kernel    operator fragments (MFMA A operand, cells padded to
//                       whole k-steps) and the LDS row offset of every (k-step,
//                       lane group)
//   item_fused2_kernel  B operand = LDS row gathered through the offset table,
//                       z-scored on the fly (one FMA), MFMA against the
//                       operator fragments streamed from L2 through a register
//                       ring.  One wave = one 16-row tile of latent variables x
//                       NT 16-voxel tiles; waves never synchronise.  Epilogue
//                       per item: shifted bootstrap moments in registers
//                       (bootstrap_permutation.py:620-626, :695), VS^T for the
//                       latent kernel, and the squared row norms of VS (:623).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "plsr_project.hip.h"

namespace plsr {

constexpr int FZ_MAXCELL = 64;      // cells of a caller
constexpr int FZ_MAXPIECE = 80;     // pieces of at most FZ_CELL_STEPS k-steps they are cut into (nz <= 320: <= 64 + 10)
constexpr int FZ_CELL_STEPS = 8;   // k-steps per cell of item_fused2_kernel (longer cells are split by the host)
constexpr int STATS_WAVES = 8;  // waves per workgroup of the statistics kernel
constexpr int STATS_REG_ROWS = 32;   // cells up to this many rows take the statistics kernel's register path

struct FusedCells {
  int32_t ncell, nkp;                  // cells, padded k-steps per item
  int32_t row_lo[FZ_MAXPIECE + 1];      // output-row range of every cell
  int32_t step_lo[FZ_MAXPIECE + 1];     // k-step range of every cell
  int32_t z[FZ_MAXPIECE];              // 1 = z-score, 0 = copy
  int32_t stat[FZ_MAXPIECE];           // cell of the scale / shift arrays (pieces of one cell share it)
};

// ---------------------------------------------------------------------------
struct StatsArgs {
  const double *X;
  int64_t ldx, p;
  int32_t n, nz, items;
  const int32_t *src;                  // [items][nz]
  FusedCells cells;
  double *sc, *sh;                     // [items][ncell][p]
};

__global__ __launch_bounds__(512) void item_stats_kernel(StatsArgs A) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t v0 = (int64_t)blockIdx.x * TV;
  const int64_t v = v0 + lane;
  // eight waves per workgroup (STATS_WAVES): the kernel is bound by the latency of dependent
  // fp64 arithmetic, and the X tile allows two workgroups per CU only -- four waves per SIMD
  const int NW = blockDim.x >> 6;
  for (int row = wave; row < A.n; row += NW)
    smem[row * TV + lane] = v < A.p ? A.X[(int64_t)row * A.ldx + v] : 0.0;
  __syncthreads();
  const int per = (A.items + gridDim.y - 1) / gridDim.y;
  const int it_lo = blockIdx.y * per;
  const int it_hi = min(A.items, it_lo + per);
  const int ncell = A.cells.ncell;
  // (item, cell) pairs round-robin over the four waves
  // source rows of a pair's cell (lane r: row lo + r), fetched one pair ahead so that the
  // global load's latency passes under the previous pair's arithmetic
  auto cell_rows = [&](int e2) {
    if (e2 >= it_hi * ncell) return 0;
    const int item2 = e2 / ncell;
    const int c2 = e2 - item2 * ncell;
    const int lo2 = A.cells.row_lo[c2], hi2 = A.cells.row_lo[c2 + 1];
    return lane < hi2 - lo2 ? A.src[(int64_t)item2 * A.nz + lo2 + lane] * TV : 0;
  };
  int mine_next = cell_rows(it_lo * ncell + wave);
  for (int e = it_lo * ncell + wave; e < it_hi * ncell; e += NW) {
    const int mine0 = mine_next;
    mine_next = cell_rows(e + NW);
    const int item = e / ncell;
    const int c = e - item * ncell;
    const int lo = A.cells.row_lo[c], hi = A.cells.row_lo[c + 1];
    double sc = 1.0, sh = 0.0;
    if (A.cells.z[c]) {
      const int32_t *src = A.src + (int64_t)item * A.nz;
      const double cnt = (double)(hi - lo);
      // the cell's source rows, 64 at a time in one vector register (lane r holds
      // row lo + r); v_readlane feeds the LDS address, so the row loop carries no
      // memory latency of its own
      // (four independent partial sums: four LDS reads in flight per wave)
      auto at = [&](int mine, int r) { return smem[__builtin_amdgcn_readlane(mine, r) + lane]; };
      double mu, var;
      // a cell of at most 32 rows is read from LDS ONCE, every read in flight at the same
      // time, and both passes run on registers (the two-pass loops below keep four reads
      // in flight and read every row twice: latency-bound).  Three sizes, so that a
      // ten-row cell does not pay for thirty-two guarded slots.
      auto reg_path = [&](auto cap) {
        constexpr int N = decltype(cap)::value;
        const int mine = mine0;
        const int m = hi - lo;
        double x[N];
#pragma unroll
        for (int r = 0; r < N; ++r) x[r] = r < m ? at(mine, r) : 0.0;
        double m0 = 0.0, m1 = 0.0, m2 = 0.0, m3 = 0.0;
#pragma unroll
        for (int r = 0; r < N; r += 4) {
          m0 += x[r];
          m1 += x[r + 1];
          m2 += x[r + 2];
          m3 += x[r + 3];
        }
        mu = ((m0 + m1) + (m2 + m3)) / cnt;
        double q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
#pragma unroll
        for (int r = 0; r < N; r += 4) {
          // (rows past the cell hold 0: their deviation is masked, not subtracted)
          const double d0 = r < m ? x[r] - mu : 0.0, d1 = r + 1 < m ? x[r + 1] - mu : 0.0;
          const double d2 = r + 2 < m ? x[r + 2] - mu : 0.0, d3 = r + 3 < m ? x[r + 3] - mu : 0.0;
          q0 = fma(d0, d0, q0);
          q1 = fma(d1, d1, q1);
          q2 = fma(d2, d2, q2);
          q3 = fma(d3, d3, q3);
        }
        var = (q0 + q1) + (q2 + q3);
      };
      if (hi - lo <= 8) {
        reg_path(std::integral_constant<int, 8>{});
      } else if (hi - lo <= 16) {
        reg_path(std::integral_constant<int, 16>{});
      } else if (hi - lo <= STATS_REG_ROWS) {
        reg_path(std::integral_constant<int, STATS_REG_ROWS>{});
      } else {
      double m0 = 0.0, m1 = 0.0, m2 = 0.0, m3 = 0.0;
      for (int r0 = lo; r0 < hi; r0 += 64) {
        const int mine = r0 + lane < hi ? src[r0 + lane] * TV : 0;
        const int m = min(64, hi - r0);
        int r = 0;
        for (; r + 4 <= m; r += 4) {
          m0 += at(mine, r);
          m1 += at(mine, r + 1);
          m2 += at(mine, r + 2);
          m3 += at(mine, r + 3);
        }
        for (; r < m; ++r) m0 += at(mine, r);
      }
      mu = ((m0 + m1) + (m2 + m3)) / cnt;
      double q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
      for (int r0 = lo; r0 < hi; r0 += 64) {
        const int mine = r0 + lane < hi ? src[r0 + lane] * TV : 0;
        const int m = min(64, hi - r0);
        int r = 0;
        for (; r + 4 <= m; r += 4) {
          const double d0 = at(mine, r) - mu, d1 = at(mine, r + 1) - mu;
          const double d2 = at(mine, r + 2) - mu, d3 = at(mine, r + 3) - mu;
          q0 = fma(d0, d0, q0);
          q1 = fma(d1, d1, q1);
          q2 = fma(d2, d2, q2);
          q3 = fma(d3, d3, q3);
        }
        for (; r < m; ++r) {
          const double d = at(mine, r) - mu;
          q0 = fma(d, d, q0);
        }
      }
      var = (q0 + q1) + (q2 + q3);
      }
      // scipy.stats.zscore's constant-slice rule (sd <= eps |mu|) followed by nan_to_num -> 0,
      // and sc = 1 / (sd sqrt(n_c)) with sd = sqrt(var / n_c), i.e. 1 / sqrt(var): one rsqrt
      // instead of a division, two square roots and another division (this kernel is bound by
      // the latency of its dependent fp64 arithmetic)
      const double em = 2.220446049250313e-16 * fabs(mu);
      const bool dead = !(var > cnt * em * em);
      sc = dead ? 0.0 : rsqrt(var);
      sh = dead ? 0.0 : -mu * sc;
    }
    if (v < A.p) {
      A.sc[(int64_t)e * A.p + v] = sc;
      A.sh[(int64_t)e * A.p + v] = sh;
    }
  }
}

// ---------------------------------------------------------------------------
struct MetaArgs {
  const double *rows;                  // [items][k][nz] operator rows
  const int32_t *src;                  // [items][nz]
  int32_t items, k, nz, MC;
  int32_t row_bytes;                   // bytes of one row of the fused kernel's LDS tile
  FusedCells cells;
  double *frag;                        // [MC][items][nkp][64]
  int32_t *rowoff;                     // [items][nkp][4]  byte offset of the LDS row
};

__global__ __launch_bounds__(256) void item_meta_kernel(MetaArgs A) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int nkp = A.cells.nkp;
  const int64_t total = (int64_t)A.MC * A.items * nkp * 64;
  if (e >= total) return;
  const int lane = (int)(e & 63);
  const int s = (int)((e >> 6) % nkp);
  const int item = (int)(((e >> 6) / nkp) % A.items);
  const int mc = (int)((e >> 6) / ((int64_t)nkp * A.items));
  int c = 0;
  while (c + 1 < A.cells.ncell && s >= A.cells.step_lo[c + 1]) ++c;
  const int g = lane >> 4;
  const int i = A.cells.row_lo[c] + 4 * (s - A.cells.step_lo[c]) + g;
  const bool valid = i < A.cells.row_lo[c + 1];
  const int j = mc * 16 + (lane & 15);
  A.frag[e] = (valid && j < A.k) ? A.rows[((int64_t)item * A.k + j) * A.nz + i] : 0.0;
  if (mc == 0 && (lane & 15) == 0) {
    // padding rows point at the cell's first row (their operator entry is 0)
    const int32_t row = A.src[(int64_t)item * A.nz + (valid ? i : A.cells.row_lo[c])];
    A.rowoff[((int64_t)item * nkp + s) * 4 + g] = row * A.row_bytes;
  }
}

// ---------------------------------------------------------------------------
struct FusedArgs {
  const double *X;
  int64_t ldx, p;
  int32_t n, items, k, MC;
  FusedCells cells;
  const double *frag;                  // [MC][items][nkp][64] (+ 8 k-steps of padding)
  const int32_t *rowoff;               // [items][nkp][4]      (+ 8 k-steps of padding)
  const double *sc, *sh;               // [items][nstat][p]
  int32_t nstat;                       // cells of the statistics (>= 1; the kernel's cells may be pieces of them)
  const double *ref;                   // [p][k] shift of the moment sums, or null
  double *S1, *S2;                     // [split][p][k] partial sums (overwritten), or null
  double *vst;                         // [items][k][ldv] VS^T, or null
  int64_t ldv;
  double *rowsq_part;                  // [nvt * VB][items][MC*16] or null
  int32_t flat;                        // 1: four waves share the MC * items (tile, item) tasks evenly (MC = 3)
};

// TVX = voxels per workgroup (LDS tile n x TVX), NT = 16-voxel tiles per wave;
// the workgroup has MC * VB waves, VB = TVX / (16 NT) voxel blocks.
// Flat mode (MC = 3, NT = 4): three waves on four SIMDs leave a CU's matrix cores
// a quarter idle, so FOUR waves split the 3 * items (tile, item) tasks evenly in
// tile-major order: a wave then works on one tile for a run of items and possibly
// on the next tile for another run ("segments"); a tile's moment sums come from
// two waves and go to two partial slabs (zero-filled by the host, merged later).
//
// item_fused2_kernel: the operator fragments of a WHOLE CELL are prefetched one cell ahead
// (fc / fn, static register indices).  Its predecessor streamed them through a 4-deep ring
// whose slot phase moved from cell to cell; the four loop variants that made the phase static
// were joined by the compiler with s_waitcnt vmcnt(0) and some forty register moves at every
// cell boundary (a quarter of a five-step cell).  Cells have at most FZ_CELL_STEPS k-steps
// (the host splits longer ones; pieces share the cell's scale / shift through cells.stat).
// TVX = voxels per workgroup (LDS tile n x TVX), NT = 16-voxel tiles per wave;
// the workgroup has MC * VB waves, VB = TVX / (16 NT) voxel blocks.
// Flat mode (MC = 3, NT = 4): three waves on four SIMDs leave a CU's matrix cores
// a quarter idle, so FOUR waves split the 3 * items (tile, item) tasks evenly in
// tile-major order: a wave then works on one tile for a run of items and possibly
// on the next tile for another run ("segments"); a tile's moment sums come from
// two waves and go to two partial slabs (zero-filled by the host, merged later).
template <int NT, int TVX>
__global__ __launch_bounds__(512) void item_fused2_kernel(FusedArgs A) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int VB = TVX / (16 * NT);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int vb = A.flat ? 0 : wave / A.MC;
  const int col = lane & 15;
  const int g = lane >> 4;
  const int64_t v0 = (int64_t)blockIdx.x * TVX;

  // ---- the whole X[:, tile] stays in LDS for the life of the workgroup ----
  for (int e = tid; e < A.n * TVX; e += blockDim.x) {
    const int row = e / TVX;
    const int64_t v = v0 + (e % TVX);
    smem[e] = v < A.p ? A.X[(int64_t)row * A.ldx + v] : 0.0;
  }
  const char *Xb = (const char *)smem + ((vb * NT) * 16 + col) * 8;   // + rowoff + nt*128

  const int per = (A.items + gridDim.y - 1) / gridDim.y;
  const int it_lo = blockIdx.y * per;
  const int it_hi = min(A.items, it_lo + per);
  const int nkp = A.cells.nkp;
  const int ncell = A.cells.ncell;
  // ---- and the row-offset table of this workgroup's items (a second vector
  // stream from L2 next to the fragments stalled the vector L1) ----
  int32_t *tbl = (int32_t *)(smem + (size_t)A.n * TVX);
  {
    const int cnt = (max(it_hi - it_lo, 0) * nkp + 8) * 4;        // + 8 steps of look-ahead padding
    const int32_t *src = A.rowoff + (size_t)it_lo * nkp * 4;      // (the buffer carries the same padding)
    for (int e = tid; e < cnt; e += blockDim.x) tbl[e] = src[e];
  }
  __syncthreads();
  if (it_lo >= it_hi) return;

  // voxel of tile nt = vbase + 16 nt (kept as one register, not NT of them)
  const int64_t vbase = v0 + vb * NT * 16 + col;
  struct {
    int64_t b;
    __device__ int64_t operator[](int nt) const { return b + 16 * nt; }
  } vox{vbase};

  // plain sums over this workgroup's items; the shift by the observed VS is
  // applied when the partials are merged (moment_unshift_kernel) -- keeping the
  // shift in registers here would push the NT = 4 instance into scratch
  double s1[NT][4], s2[NT][4];
  const bool moments = A.S1 != nullptr;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s1[nt][r] = 0.0;
      s2[nt][r] = 0.0;
    }

  // this wave's segments: (tile of latent variables, run of items)
  const int cnt = it_hi - it_lo;
  int seg_mc[2], seg_a[2], seg_b[2], nseg = 1;
  if (!A.flat) {
    seg_mc[0] = wave % A.MC;
    seg_a[0] = it_lo;
    seg_b[0] = it_hi;
  } else {
    const int nw = blockDim.x >> 6;
    const int U = A.MC * cnt;
    const int u0 = (int)((int64_t)wave * U / nw), u1 = (int)((int64_t)(wave + 1) * U / nw);
    const int m0 = u0 / cnt;
    seg_mc[0] = m0;
    seg_a[0] = it_lo + (u0 - m0 * cnt);
    seg_b[0] = it_lo + min(cnt, u1 - m0 * cnt);
    if (u1 > (m0 + 1) * cnt) {
      nseg = 2;
      seg_mc[1] = m0 + 1;
      seg_a[1] = it_lo;
      seg_b[1] = it_lo + (u1 - (m0 + 1) * cnt);
    }
    if (u0 >= u1) nseg = 0;
  }

  for (int sg = 0; sg < nseg; ++sg) {
  const int mc = seg_mc[sg];
  const int sa = seg_a[sg], sb = seg_b[sg];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s1[nt][r] = 0.0;
      s2[nt][r] = 0.0;
    }

  // Stream of k-steps over (item, cell, step): fragments and row offsets are
  // contiguous per wave, so the 4-deep register rings run on across cell and
  // item boundaries.  The k-loops hold no memory operation besides the rings
  // (the scale / shift of a cell are prefetched one cell ahead, between loops).
  const double *fp = A.frag + ((size_t)mc * A.items * nkp + (size_t)it_lo * nkp) * 64 + lane;
  const int32_t *rp = tbl + g;           // LDS: byte offset of the row of (step, lane group)
  int64_t pos = (int64_t)(sa - it_lo) * nkp;    // stream position of the current cell's first step
  double fc[FZ_CELL_STEPS], fn[FZ_CELL_STEPS];
#pragma unroll
  for (int u = 0; u < FZ_CELL_STEPS; ++u) fc[u] = fp[(size_t)(pos + u) * 64];
  double bn[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bn[nt] = *(const double *)(Xb + rp[pos * 4] + nt * 128);
  int ro1 = rp[(pos + 1) * 4];           // rows of the next step

  // Unconditional loads (clamped addresses): a select on the loaded value would
  // make the prefetch wait for its own data.  Lanes past p read voxel p-1 and
  // are masked where results leave the kernel.
  auto load_cell = [&](int item, int c, double (&sc)[NT], double (&sh)[NT]) {
    const int64_t base = ((int64_t)min(item, it_hi - 1) * A.nstat + A.cells.stat[c]) * A.p;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int64_t vc = min(vox[nt], A.p - 1);
      sc[nt] = A.sc[base + vc];
      sh[nt] = A.sh[base + vc];
    }
  };

  f64x4 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = (f64x4){0.0, 0.0, 0.0, 0.0};

  int item = sa, c = 0;
  auto item_done = [&]() {
    // ---- item done: acc[nt][r] = VS[j = 16 mc + g + 4 r][voxel nt] ----
    double q[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = mc * 16 + g + 4 * r;
        const double val = acc[nt][r];
        if (moments) {
          s1[nt][r] += val;
          s2[nt][r] = fma(val, val, s2[nt][r]);
        }
        if (vox[nt] < A.p) q[r] = fma(val, val, q[r]);
        if (A.vst != nullptr && j < A.k && vox[nt] < A.p)
          A.vst[((int64_t)item * A.k + j) * A.ldv + vox[nt]] = val;
        acc[nt][r] = 0.0;
      }
    if (A.rowsq_part != nullptr) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double x = q[r];
        x += __shfl_xor(x, 1);
        x += __shfl_xor(x, 2);
        x += __shfl_xor(x, 4);
        x += __shfl_xor(x, 8);
        if (col == 0)
          A.rowsq_part[(((int64_t)blockIdx.x * VB + vb) * A.items + item) * (A.MC * 16) + mc * 16 + g + 4 * r] = x;
      }
    }
  };
  // One cell: its fragments (fcur) and scale / shift (scc / shc) were loaded during the
  // previous cell; it loads the next cell's into the OTHER set (fnxt, scx / shx).  Two sets in
  // turn (the loop below is unrolled by two) keep every register index static without
  // moves or a wait for the prefetch at the cell boundary.
  auto do_cell = [&](double (&fcur)[FZ_CELL_STEPS], double (&fnxt)[FZ_CELL_STEPS], double (&scc)[NT],
                     double (&shc)[NT], double (&scx)[NT], double (&shx)[NT]) {
    const bool last = c + 1 == ncell;
    load_cell(last ? item + 1 : item, last ? 0 : c + 1, scx, shx);
    const int ns = A.cells.step_lo[c + 1] - A.cells.step_lo[c];
    const double *fq = fp + (size_t)pos * 64;
    const int32_t *rq = rp + pos * 4;
    // the next cell's fragments (unconditional: past the last cell they read the
    // stream's padding), in flight for the whole of this cell
#pragma unroll
    for (int u = 0; u < FZ_CELL_STEPS; ++u) fnxt[u] = fq[(size_t)(ns + u) * 64];
    // one k-step: reads the raw rows of the next step and the row offsets of the one
    // after (both LDS); no memory operation besides
#pragma unroll
    for (int u = 0; u < FZ_CELL_STEPS; ++u) {
      if (u < ns) {
        double z[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) z[nt] = fma(bn[nt], scc[nt], shc[nt]);
        const int ron = ro1;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bn[nt] = *(const double *)(Xb + ron + nt * 128);
        ro1 = rq[(u + 2) * 4];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma_f64(fcur[u], z[nt], acc[nt]);
      }
    }
    pos += ns;
    if (last) {
      item_done();
      ++item;
      c = 0;
    } else {
      ++c;
    }
  };
  double scA[NT], shA[NT], scB[NT], shB[NT];
  load_cell(sa, 0, scA, shA);
  while (item < sb) {
    do_cell(fc, fn, scA, shA, scB, shB);
    if (item >= sb) break;
    do_cell(fn, fc, scB, shB, scA, shA);
  }

  if (moments) {
    // flat mode: the wave that starts a tile's items writes slab 0, the one that ends them slab 1
    const int slab = A.flat ? 2 * blockIdx.y + (sa == it_lo ? 0 : 1) : blockIdx.y;
    double *o1 = A.S1 + (int64_t)slab * A.p * A.k;
    double *o2 = A.S2 + (int64_t)slab * A.p * A.k;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = mc * 16 + g + 4 * r;
        if (j < A.k && vox[nt] < A.p) {
          o1[vox[nt] * A.k + j] = s1[nt][r];
          o2[vox[nt] * A.k + j] = s2[nt][r];
        }
      }
  }
  }   // segments
}

// Multiblock operator rows on the device (class_functions.py:503-505 + :620):
//   out[b][j][i] = sum_r U[r][j] / sqrt(rowsq[b][r]) * raw[b][r][i]
// i.e. the un-normalised multiblock rows of resample b, each scaled by the inverse of
// its norm over all voxels (a zero row stays zero, as _normalize does) and projected on
// U -- formed where the norms already are, so the host never waits for them.
__global__ __launch_bounds__(256) void scale_project_rows_kernel(const double *raw, const double *rowsq,
                                                                int64_t rowsq_stride, const double *U, int items,
                                                                int kr, int nz, int k, double *out) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)items * k * nz) return;
  const int i = (int)(e % nz);
  const int j = (int)((e / nz) % k);
  const int b = (int)(e / ((int64_t)nz * k));
  const double *rb = raw + (int64_t)b * kr * nz + i;
  const double *qb = rowsq + (int64_t)b * rowsq_stride;
  double a = 0.0;
  for (int r = 0; r < kr; ++r) {
    const double q = qb[r];
    const double inv = q > 0.0 ? 1.0 / sqrt(q) : 0.0;
    a = fma(U[(int64_t)r * k + j] * inv, rb[(int64_t)r * nz], a);
  }
  out[e] = a;
}

// S1 += sum_b (x_b - ref),  S2 += sum_b (x_b - ref)^2  from the plain partial
// sums  P1 = sum x_b,  P2 = sum x_b^2  of `items` resamples (fixed split order).
// The expansion loses eps * mean^2 / var relative accuracy, the same as the
// final variance formula (boot_finalize_kernel) does anyway.
__global__ __launch_bounds__(256) void moment_unshift_kernel(double *S1, double *S2, const double *P1,
                                                            const double *P2, const double *ref,
                                                            int64_t count, int nsplit, double items) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= count) return;
  double a = P1[e], b = P2[e];
  for (int c = 1; c < nsplit; ++c) {
    a += P1[(int64_t)c * count + e];
    b += P2[(int64_t)c * count + e];
  }
  const double r = ref != nullptr ? ref[e] : 0.0;
  S1[e] += a - items * r;
  S2[e] += fma(r, fma(items, r, -2.0 * a), b);
}

}  // namespace plsr
