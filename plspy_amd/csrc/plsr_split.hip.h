// K2s: per-split Grams of behaviour / multiblock PLS in two stages, one wave per (item, voxel chunk).
//
// A split-half item (split_half_resampling.py:119-197, :266-383, :548-683, :687-802) stacks the two
// halves' cross-blocks M = [M1; M2]; everything the reference derives from the halves' SVDs follows from
// G = M M^T (k x k blocks G11, G12, G22).  A half's cross-block is, per (group, condition) CELL of the
// half's rows (class_functions.py:185-247, :454-516):
//
//   behaviour rows   R_c[beh, v] = sum_{r in c} Yz_c[r, beh] (X[r, v] - mean_c(v)) / sqrt(ss_c(v))
//                                = sc_c(v) * sum_{r in c} Yz_c[r, beh] X[r, v]         (Yz_c sums to 0 over c)
//   task rows        T[j, v]     = sum_c Wc[j, c] * S_c(v),   S_c(v) = sum_{r in c} X[r, v]
//                                  (the mean-centring operator is constant inside a cell)
//
// so the item needs, per voxel: the cells' sums and centred sums of squares (two-pass, from the rows in
// registers), one small product per behaviour cell on the RAW rows, a scale of its accumulator, one tiny
// product for the task rows, and the Gram of the lot.  The fused kernel of round 2
// (gram_kernel<.., FUSED>, plsr_gram.hip.h) staged every row z-scored through LDS and multiplied the
// stacked dense operator (200 rows x 76 at config 4): 150 + 60 MFMAs per 16 voxels behind sixty vector
// loads, twenty parked rows and two barriers per K-chunk -- matrix pipe 30 % busy.  Here:
//
//   * orientation D[voxel][operator row]: the gathered rows of X are the MFMA A operand, loaded
//     STRAIGHT from global memory into that layout (lane (voxel, kk) reads row src[4 s + kk] of its
//     voxel: one 512-byte access of four 128-byte row segments, scalar base + one 32-bit lane offset
//     per slot, kept in registers for the whole chunk) -- no LDS staging, no barrier, no gather kernel;
//   * a ring of one register pair per (cell, k-step) slot: slot j of tile t + 1 is requested right
//     after slot j of tile t has been consumed, so every load has a whole tile's MFMAs to arrive;
//   * statistics on the same registers: the lane's own rows are summed on the VALU, the four kk lanes
//     of a voxel are summed by v_mfma_f64_4x4x4 (A = ones: every lane gets its voxel's cell sum; A =
//     one-hot: lane (i, voxel) collects cell 4 g + i) -- which is also the layout of the task product's
//     A operand and, through a 16-entry LDS patch, of the accumulators' scale;
//   * the scaled accumulators are, unchanged, both operands of the Gram MFMAs (as in gram_kernel);
//   * 24 + 3 + 60 MFMAs per 16 voxels at config 4 (k = 38) instead of 210.
//
// One wave = one workgroup = (item, voxel chunk); G lives in registers for the chunk (15 tiles = 120
// VGPRs at five row tiles) and leaves as one partial per chunk; split_reduce_kernel sums the chunks in
// fixed order, undoes the kernel's row order and applies the multiblock row normalisation
// (class_functions.py:503-505 on the Gram: G_ij / (|row_i| |row_j|)).  Workgroup ids go round the
// eight XCDs, and id mod 8 picks the voxel range: the waves of an XCD sweep the same eighth of X, which
// then crosses the fabric once per XCD instead of once per item.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "plsr_agg.hip.h"

namespace plsr {

constexpr int SG_MAXQ = 20;          // cell slots of an item (behaviour cells first, then task-only cells)
constexpr int SG_BP = 8;             // behaviour rows per cell in a 16-column tile (b <= 8)
constexpr int SG_IP = 16 / SG_BP;    // cells per tile

struct SplitMetaArgs {
  const int32_t *xsrc, *ysrc;        // [items][nz] rows of X / of Y per (cell, row), cells concatenated in slot order
  const double *Y;                   // [ny][b]
  int32_t b, items, nz, nq, nbq, cs; // nq cell slots of cs k-steps; the first nbq carry behaviour rows
  int32_t cell_lo[SG_MAXQ + 1];
  int64_t ldx_bytes;
  uint32_t *roff;                    // [items][nq * cs][4] byte offset of the slot's source row in X
  double *bfrag;                     // [items][nbq * cs][64] stage-1 B fragments (Yz, zero padded)
};

// roff: slot (q, s), lane group kk -> row xsrc[cell_lo[q] + 4 s + kk] (rows past the cell repeat its first
// row: loaded, never used).  bfrag: lane (col, kk) of slot (q, s) = Yz_q[4 s + kk][col - (q % IP) BP] with
// Yz_q the cell's rows of Y z-scored per column (ddof 0, / sqrt(n_c), constant columns -> 0:
// class_functions.py:229-238).
__global__ __launch_bounds__(256) void split_meta_kernel(SplitMetaArgs A) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t n1 = (int64_t)A.items * A.nq * A.cs * 4;
  const int64_t n2 = (int64_t)A.items * A.nbq * A.cs * 64;
  if (e < n1) {
    const int kk = (int)(e & 3);
    int64_t t = e >> 2;
    const int s = (int)(t % A.cs);
    t /= A.cs;
    const int q = (int)(t % A.nq);
    const int item = (int)(t / A.nq);
    const int nc = A.cell_lo[q + 1] - A.cell_lo[q];
    const int r = 4 * s + kk;
    const int32_t row = nc > 0 ? A.xsrc[(int64_t)item * A.nz + A.cell_lo[q] + (r < nc ? r : 0)] : 0;
    A.roff[e] = (uint32_t)((int64_t)row * A.ldx_bytes);
  } else if (e - n1 < n2) {
    const int64_t f = e - n1;
    const int lane = (int)(f & 63);
    int64_t t = f >> 6;
    const int s = (int)(t % A.cs);
    t /= A.cs;
    const int q = (int)(t % A.nbq);
    const int item = (int)(t / A.nbq);
    const int col = lane & 15, kk = lane >> 4;
    const int beh = col - (q % SG_IP) * SG_BP;
    const int lo = A.cell_lo[q], nc = A.cell_lo[q + 1] - lo;
    const int r = 4 * s + kk;
    double z = 0.0;
    if (beh >= 0 && beh < A.b && beh < SG_BP && r < nc) {
      const int32_t *ys = A.ysrc + (int64_t)item * A.nz + lo;
      double mu = 0.0;
      for (int i = 0; i < nc; ++i) mu += A.Y[(int64_t)ys[i] * A.b + beh];
      mu /= (double)nc;
      double ss = 0.0;
      for (int i = 0; i < nc; ++i) {
        const double d = A.Y[(int64_t)ys[i] * A.b + beh] - mu;
        ss = fma(d, d, ss);
      }
      const double sd = sqrt(ss / (double)nc);
      if (sd > 2.220446049250313e-16 * fabs(mu)) z = (A.Y[(int64_t)ys[r] * A.b + beh] - mu) / sd / sqrt((double)nc);
    }
    A.bfrag[f] = z;
  }
}

// task coefficients as B fragments: group g4 of four cells, lane (col = task row j, kk) = Wc[j][4 g4 + kk]
__global__ __launch_bounds__(64) void split_wfrag_kernel(const double *Wc, int ktask, int nq, double *wfrag) {
  const int g4 = blockIdx.x, lane = threadIdx.x;
  const int j = lane & 15, c = 4 * g4 + (lane >> 4);
  wfrag[g4 * 64 + lane] = (j < ktask && c < nq) ? Wc[j * nq + c] : 0.0;
}

struct SplitArgs {
  const double *X;
  int64_t p;
  int32_t items, nq, nbq, cs;          // run-time counts (the template's are maxima unless EXACT)
  int32_t nrow[SG_MAXQ];               // rows of every cell slot (0: empty slot)
  double rn[SG_MAXQ];                  // 1 / rows (0 for an empty slot)
  const uint32_t *roff;                // [items][nq * cs][4]
  const double *bfrag;                 // [items][nbq * cs][64]
  const double *wfrag;                 // [ceil(nq / 4)][64] or null
  int32_t nx, csub;                    // voxel ranges per item: nx (<= 8, the XCDs) x csub
  int64_t ntile;                       // 16-voxel tiles
  double *Gp;                          // [items][nx * csub][NG][4][64]
  // ROWS variant (split_rows: the rows themselves instead of their Gram)
  double *R;                           // [items][m][ldv] the scaled rows, logical order
  int64_t ldv;
  int32_t m;
  int16_t rowof[7 * 16];               // the kernel's row (tile * 16 + column) -> logical row, or -1
  double *rowsq_part;                  // [items][nx * csub][MC * 16] squared norms of the kernel's rows over the range
};


// 1 / sqrt(x) for a normal, positive x without the library routine's range checks (they came out as divergent
// branches in the middle of the MFMA stream): v_rsq_f64 seeds about 26 bits, the second-order correction
// y0 + y0 e (1/2 + 3/8 e), e = 1 - x y0^2, leaves an error of e^3
__device__ __forceinline__ double rsqrt_pos(double x) {
  const double y0 = __builtin_amdgcn_rsq(x);
  const double e = fma(-(x * y0), y0, 1.0);
  return fma(y0 * e, fma(e, 0.375, 0.5), y0);
}

constexpr int split_gram_m1(int MC, int idx) {
  int m1 = 0, base = 0;
  while (idx >= base + (MC - m1)) {
    base += MC - m1;
    ++m1;
  }
  return m1;
}
constexpr int split_gram_m2(int MC, int idx) {
  int m1 = 0, base = 0;
  while (idx >= base + (MC - m1)) {
    base += MC - m1;
    ++m1;
  }
  return m1 + (idx - base);
}

// NTB behaviour tiles (two cells each), NTO pairs of task-only cells, NTT (0 / 1) task tile, CS k-steps per
// cell.  EXACT: the launch has exactly these counts and every cell has more than 4 (CS - 1) rows, so all
// guards fold and only a cell's last k-step is masked; the item's stage-1 fragments then live in registers
// (the generic instances read them from LDS).
//
// The tile loop is software-pipelined by hand: iteration t forms the scaled rows D of tile t (statistics, stage
// 1, scale) and, cell by cell between those steps, issues the Gram MFMAs of tile t - 1's rows -- sixty
// independent MFMAs that fill the matrix pipe while the wave walks the dependent chain sum -> 4x4x4 MFMA ->
// mean -> centred squares -> 4x4x4 MFMA -> rsqrt of the next tile (with one wave per SIMD nothing else would).
// Scheduling barriers between the cells keep that interleave.
// ROWS: the rows themselves are the result (stored in logical order, with their squared norms over the range) and no
// Gram is formed -- the first pass of the multiblock bootstrap (engine.split_rows -> plsr_rows_project).
template <int NTB, int NTO, int NTT, int CS, bool EXACT, bool ROWS = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(ROWS && EXACT ? 2 : 1)))
void split_gram_kernel(SplitArgs A) {
  constexpr int IP = SG_IP;
  constexpr int NP = NTB + NTO;                    // cell pairs
  constexpr int NQ = NP * IP;                      // cell slots
  constexpr int NS = NQ * CS;                      // ring slots
  constexpr int MC = NTB + NTT;                    // row tiles of the Gram
  constexpr int NG = MC * (MC + 1) / 2;
  constexpr int NG4 = (NQ + 3) / 4;
  constexpr int NGM = 4 * NG;                      // Gram MFMAs per tile
  constexpr int GPC = (NGM + NQ - 1) / NQ;         // ... issued per cell of the next tile
  constexpr bool BFREG = EXACT;
  constexpr int NBF = NTB * IP * CS;
  static_assert(NQ <= SG_MAXQ && NS <= 63, "cell slots / outstanding loads");
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x;
  const int col = lane & 15, kk = lane >> 4;
  const int nq = EXACT ? NQ : A.nq, nbq = EXACT ? NTB * IP : A.nbq, cs = EXACT ? CS : A.cs;

  // ---- which (item, voxel range) ----
  const int64_t wg = blockIdx.x;
  const int xcd = (int)(wg % A.nx);
  const int64_t rest = wg / A.nx;
  const int item = (int)(rest % A.items);
  const int sub = (int)(rest / A.items);
  const int64_t per_x = (A.ntile + A.nx - 1) / A.nx;
  const int64_t per_s = (per_x + A.csub - 1) / A.csub;
  const int64_t x_lo = (int64_t)xcd * per_x, x_hi = min(A.ntile, x_lo + per_x);
  const int64_t t_lo = min(x_hi, x_lo + (int64_t)sub * per_s), t_hi = min(x_hi, t_lo + per_s);
  const int chunk = xcd * A.csub + sub;

  // ---- LDS: per-slot row offsets (+ this lane's voxel), [slot][lane] -- sixty of them in registers beside
  // the ring's sixty pairs and the Gram's 120 accumulators left the config-4 instance 80 bytes of scratch;
  // then (generic instances) this item's stage-1 fragments ----
  uint32_t *rl = (uint32_t *)smem;                  // [NS][64]
  double *bf = smem + NS * 32;                      // [nbq * cs][64]
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int s = 0; s < CS; ++s) {
      const bool live = EXACT || (q < nq && s < cs);
      rl[(q * CS + s) * 64 + lane] =
          A.roff[((size_t)item * nq * cs + (live ? q * cs + s : 0)) * 4 + kk] + (uint32_t)col * 8u;
    }
  double bfr[BFREG ? NBF : 1];
  {
    const double *src = A.bfrag + (size_t)item * nbq * cs * 64;
    if (BFREG) {
#pragma unroll
      for (int e = 0; e < NBF; ++e) bfr[e] = src[e * 64 + lane];
    } else {
      for (int e = 0; e < nbq * cs; ++e) bf[e * 64 + lane] = src[e * 64 + lane];
    }
  }
  asm volatile("" ::: "memory");
  // per-lane constants of the finalising layout: lane (i = kk, voxel = col) owns cell 4 g4 + i
  double cntl[NG4], rnl[NG4];
#pragma unroll
  for (int g4 = 0; g4 < NG4; ++g4) {
    const int q = 4 * g4 + kk;
    const int nr = q < NQ ? A.nrow[q] : 0;
    cntl[g4] = (double)nr;
    rnl[g4] = nr > 0 ? 1.0 / (double)nr : 0.0;
  }
  double wf[NTT ? NG4 : 1];
  if (NTT) {
#pragma unroll
    for (int g4 = 0; g4 < NG4; ++g4) wf[g4] = A.wfrag[g4 * 64 + lane];
  }
  const double one = 1.0;
  double hot[4], khot[4];          // one-hot A operands of the 4x4x4 MFMA; this lane's kk as a 0 / 1 weight
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    hot[u] = (lane & 3) == u ? 1.0 : 0.0;
    khot[u] = kk == u ? 1.0 : 0.0;
  }
  // B operands that carry a group's scales from the finalising layout (A[m = voxel][k = cell-in-group]) into
  // the accumulator layout of its two behaviour tiles: B[k][n = column] = 1 where column n belongs to cell k
  double selb[2];
  selb[0] = kk == (col >> 3) ? 1.0 : 0.0;
  selb[1] = kk == 2 + (col >> 3) ? 1.0 : 0.0;

  f64x4 G[ROWS ? 1 : NG];
#pragma unroll
  for (int i = 0; i < (ROWS ? 1 : NG); ++i) G[i] = (f64x4){0.0, 0.0, 0.0, 0.0};
  // ROWS: the byte offset of every kernel row's logical row in R (-1: no such row), a transpose patch, and per
  // row tile the running sum of squares of this lane's (column's) voxels
  int64_t *rofl = (int64_t *)(smem + NS * 32 + (BFREG ? 0 : (size_t)nbq * cs * 64));
  double *tp = (double *)(rofl + (ROWS ? MC * 16 : 0));
  double rq[ROWS ? MC : 1];
  if (ROWS) {
#pragma unroll
    for (int m = 0; m < MC; ++m) {
      rq[m] = 0.0;
      if (lane < 16) {
        const int lr = A.rowof[m * 16 + lane];
        rofl[m * 16 + lane] = lr < 0 ? -1 : ((int64_t)item * A.m + lr) * A.ldv * 8;
      }
    }
  }
  // the previous tile's scaled rows (zero before the first) and the ones being formed.  (The loop body written
  // twice with the two sets' roles swapped, to save the copy at the end of a tile, spilled 700 bytes per lane:
  // the set being formed must sit in VGPRs for its scaling, the copy is what moves it to accumulator registers.)
  f64x4 Da[ROWS ? 1 : MC], Db[MC];
#pragma unroll
  for (int m = 0; m < MC; ++m) Db[m] = (f64x4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int m = 0; m < (ROWS ? 1 : MC); ++m) Da[m] = (f64x4){0.0, 0.0, 0.0, 0.0};

  // tile t covers voxels [16 t, 16 t + 16); the last one is read from p - 16 and masks what tile t - 1 had
  auto tile_v0 = [&](int64_t t) { return min(16 * t, A.p - 16); };
  const char *Xb = (const char *)A.X;
  double xr[NS];
  if (t_lo < t_hi) {
    const char *b0 = Xb + tile_v0(t_lo) * 8;
#pragma unroll
    for (int j = 0; j < NS; ++j) xr[j] = *(const double *)(b0 + rl[j * 64 + lane]);
  }

  // forms the scaled rows D of `tile` and adds the Gram of the previous tile's rows Dc
  auto body = [&](int64_t tile, const f64x4 (&Dc)[ROWS ? 1 : MC], f64x4 (&D)[MC]) __attribute__((always_inline)) {
    const int64_t tcl = min(tile, t_hi - 1);
    const int64_t v0 = tile_v0(tcl);
    const bool vvalid = v0 + col >= 16 * tcl;
    const char *bn = Xb + tile_v0(min(tile + 1, t_hi - 1)) * 8;     // next tile's rows (the last tile re-reads itself)
#pragma unroll
    for (int m = 0; m < MC; ++m) D[m] = (f64x4){0.0, 0.0, 0.0, 0.0};

#pragma unroll
    for (int g4 = 0; g4 < NG4; ++g4) {
      double totc[4] = {0.0, 0.0, 0.0, 0.0};
      double ssq = 0.0;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int q = 4 * g4 + u;
        if (q < NQ) {
          if (EXACT || q < nq) {
            const int t = q / IP;
            const int nr = A.nrow[q];
            double x[CS], xm[CS];
            // the lane's own rows of the cell (4 s + kk < n_c), summed; then over the four kk lanes
            double s1p = 0.0;
#pragma unroll
            for (int s = 0; s < CS; ++s) {
              x[s] = xr[q * CS + s];
              const bool ok = (EXACT && s < CS - 1) || 4 * s + kk < nr;
              xm[s] = ok ? x[s] : 0.0;
              s1p += xm[s];
            }
            const double tot = mfma4_f64(one, s1p, 0.0);
            totc[u] = tot;
            if (t < NTB && (EXACT || q < nbq)) {
              const double mean = tot * A.rn[q];
              double ssp = 0.0;
#pragma unroll
              for (int s = 0; s < CS; ++s) {
                const bool ok = (EXACT && s < CS - 1) || 4 * s + kk < nr;
                const double d = ok ? x[s] - mean : 0.0;
                ssp = fma(d, d, ssp);
              }
              ssq = mfma4_f64(hot[u], ssp, ssq);
#pragma unroll
              for (int s = 0; s < CS; ++s)
                if (EXACT || s < cs)
                  D[t] = mfma_f64(x[s], BFREG ? bfr[BFREG ? q * CS + s : 0] : bf[(q * cs + s) * 64 + lane], D[t]);
            }
            // the cell's slots are requested again for the next tile AFTER their last use: requested ahead
            // of it the loads needed registers of their own, and the ring was copied back -- behind an
            // s_waitcnt vmcnt(0) -- at the end of every tile
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < CS; ++s) xr[q * CS + s] = *(const double *)(bn + rl[(q * CS + s) * 64 + lane]);
          }
          // ---- this cell's share of the previous tile's Gram ----
          if (!ROWS) {
#pragma unroll
            for (int j = q * GPC; j < (q + 1) * GPC && j < NGM; ++j) {
              const int r = j / NG, idx = j % NG;
              G[ROWS ? 0 : idx] = mfma_f64(Dc[ROWS ? 0 : split_gram_m1(MC, idx)][r], Dc[ROWS ? 0 : split_gram_m2(MC, idx)][r],
                                           G[ROWS ? 0 : idx]);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // ---- the group's four cells: lane (i = kk, voxel = col) finalises cell 4 g4 + i ----
      // (a weighted sum with 0 / 1 weights: nested selects on values that live in accumulator registers
      // became divergent branches)
      const double s1sel = fma(totc[0], khot[0], fma(totc[1], khot[1], fma(totc[2], khot[2], totc[3] * khot[3])));
      if (NTT) D[MC - 1] = mfma_f64(vvalid ? s1sel : 0.0, wf[g4], D[MC - 1]);
      if (4 * g4 < NTB * IP) {
        const double mean = s1sel * rnl[g4];
        const double em = 2.220446049250313e-16 * fabs(mean);
        const bool dead = !(ssq > cntl[g4] * em * em) || !vvalid;
        const double y = rsqrt_pos(dead ? 1.0 : ssq);
        const double scv = dead ? 0.0 : y;
        // scale the group's behaviour tiles: the scales cross into the accumulator layout (lane (column, kk),
        // register r = voxel kk + 4 r) through one MFMA per tile
#pragma unroll
        for (int tt = 0; tt < 4 / IP; ++tt) {
          const int t = g4 * (4 / IP) + tt;
          if (t < NTB && (EXACT || t * IP < nbq)) {
            const f64x4 sc4 = mfma_f64(scv, selb[tt], (f64x4){0.0, 0.0, 0.0, 0.0});
#pragma unroll
            for (int r = 0; r < 4; ++r) D[t][r] *= sc4[r];
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (ROWS) {
      // The tile's rows leave.  In the accumulator a lane holds four voxels 32 bytes apart of ONE row: stored as
      // they are, every store instruction touched sixteen 128-byte lines with 32 bytes each and the launch wrote at
      // 1.9 TB/s (14 of its 32 us per item; without the stores: 18).  So a row tile goes through a per-wave LDS
      // patch (16 x 18 doubles) and leaves as whole lines: lane l stores voxels 2 (l % 8), + 1 of row l / 8 (+ 8).
#pragma unroll
      for (int m = 0; m < MC; ++m) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double val = D[m][r];
          const bool mine = v0 + kk + 4 * r >= 16 * tcl;
          if (mine) rq[m] = fma(val, val, rq[m]);
          tp[col * 18 + kk + 4 * r] = val;
        }
        asm volatile("" ::: "memory");
        typedef double d2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int row = h * 8 + (lane >> 3), vp = 2 * (lane & 7);
          const d2 pair = *(const d2 *)(tp + row * 18 + vp);
          const int64_t ro = rofl[m * 16 + row];
#if !(PLSR_ABLATE & 4096)                 // (dev: the ROWS variant without its stores)
          if (ro >= 0) {
            char *dst = (char *)A.R + ro + (v0 + vp) * 8;
            if (v0 == 16 * tcl) {
              *(d2 *)dst = pair;
            } else {                       // the shifted last tile: only the voxels tile t - 1 did not have
              if (v0 + vp >= 16 * tcl) *(double *)dst = pair.x;
              if (v0 + vp + 1 >= 16 * tcl) *(double *)(dst + 8) = pair.y;
            }
          }
#endif
        }
        asm volatile("" ::: "memory");
      }
    }
  };

  // The first tile's loads are drained before the loop: the compiler's wait-count pass merges the state at
  // the loop header with the back edge's, and with the prologue's sixty loads still pending (in whatever order
  // the scheduler issued them) it put s_waitcnt vmcnt(0) at the top of EVERY tile; drained, it waits for exactly
  // the cell's three slots (vmcnt(59), (58), (57)) and the other fifty-seven stay in flight.
  __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0)
  // one iteration more than there are tiles: the last one only has tile t_hi - 1's Gram to do (the rows it
  // forms from the re-read last tile are dropped)
  if (t_lo < t_hi) {
    for (int64_t tile = t_lo; tile <= (ROWS ? t_hi - 1 : t_hi); ++tile) {
      body(tile, Da, Db);
      if (!ROWS) {
#pragma unroll
        for (int m = 0; m < (ROWS ? 1 : MC); ++m) Da[m] = Db[ROWS ? 0 : m];
      }
    }
  }
  if (ROWS) {
    // squared norms of this lane's rows over the range: the four kk lanes of a column together
    double *o = A.rowsq_part + ((size_t)item * A.nx * A.csub + chunk) * MC * 16;
#pragma unroll
    for (int m = 0; m < MC; ++m) {
      double x = rq[m];
      x += __shfl_xor(x, 16);
      x += __shfl_xor(x, 32);
      if (kk == 0) o[m * 16 + col] = x;
    }
    return;
  }

  double *out = A.Gp + ((size_t)item * A.nx * A.csub + chunk) * NG * 256;
#pragma unroll
  for (int i = 0; i < NG; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) out[(i * 4 + r) * 64 + lane] = G[i][r];
}

// LDS of a workgroup: row offsets of the instance's ns ring slots, then (generic instances) the stage-1 fragments
inline size_t split_lds_bytes(int nbq, int cs, int ns, bool exact, int rows_mc = 0) {
  return (size_t)ns * 64 * sizeof(uint32_t) + (exact ? 0 : (size_t)nbq * cs * 64 * sizeof(double)) +
         (rows_mc ? (size_t)rows_mc * 16 * sizeof(int64_t) + 16 * 18 * sizeof(double) : 0);
}

struct SplitReduceArgs {
  const double *Gp;                    // [items][nchunk][NG][4][64]
  int32_t items, nchunk, MC;           // MC row tiles in the kernel's order
  int32_t m, mm;                       // logical rows, padded row count of d_G
  int32_t normalise;
  int16_t inv[7 * 16];                 // logical row -> the kernel's row (tile * 16 + column)
  double *G;                           // [items][mm][mm]
  double *rownorm;                     // [items][mm] or null: sqrt of the raw diagonal (the rows' norms over all voxels)
};

// one workgroup per item: chunks summed in fixed order into an LDS image of the kernel-order Gram, then
// written in logical order, optionally as G_ij / (sqrt(G_ii) sqrt(G_jj)) (0 where a row has norm 0)
__global__ __launch_bounds__(256) void split_reduce_kernel(SplitReduceArgs A) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int item = blockIdx.x;
  const int MM = A.MC * 16, LD = MM + 1;
  const int NG = A.MC * (A.MC + 1) / 2;
  const double *src = A.Gp + (size_t)item * A.nchunk * NG * 256;
  for (int e = threadIdx.x; e < NG * 256; e += 256) {
    double acc = 0.0;
    for (int c = 0; c < A.nchunk; ++c) acc += src[(size_t)c * NG * 256 + e];
    const int idx = e >> 8, r = (e >> 6) & 3, lane = e & 63;
    int m1 = 0, base = 0;
    while (idx >= base + (A.MC - m1)) {
      base += A.MC - m1;
      ++m1;
    }
    const int m2 = m1 + (idx - base);
    const int c1 = m1 * 16 + (lane >> 4) + 4 * r, c2 = m2 * 16 + (lane & 15);
    smem[c1 * LD + c2] = acc;
    if (m1 != m2) smem[c2 * LD + c1] = acc;
  }
  __syncthreads();
  double *out = A.G + (size_t)item * A.mm * A.mm;
  for (int e = threadIdx.x; e < A.mm * A.mm; e += 256) {
    const int l1 = e / A.mm, l2 = e % A.mm;
    double v = 0.0;
    if (l1 < A.m && l2 < A.m) {
      const int i1 = A.inv[l1], i2 = A.inv[l2];
      v = smem[i1 * LD + i2];
      if (A.normalise) {
        const double d1 = sqrt(smem[i1 * LD + i1]), d2 = sqrt(smem[i2 * LD + i2]);
        v = (d1 * d2 > 0.0) ? v / d1 / d2 : 0.0;
      }
    }
    out[e] = v;
  }
  if (A.rownorm != nullptr)
    for (int l = threadIdx.x; l < A.mm; l += 256)
      A.rownorm[(size_t)item * A.mm + l] = l < A.m ? sqrt(smem[A.inv[l] * LD + A.inv[l]]) : 0.0;
}

// ROWS: d_rowsq[item][l] = sum over the ranges of the squared norm of the kernel's row that is logical row l
__global__ __launch_bounds__(256) void split_rowsq_kernel(const double *part, int items, int nchunk, int MC, int m,
                                                          int64_t stride, SplitReduceArgs inv, double *rowsq) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)items * m) return;
  const int item = (int)(e / m), l = (int)(e % m);
  const int at = inv.inv[l];
  double acc = 0.0;
  for (int c = 0; c < nchunk; ++c) acc += part[((size_t)item * nchunk + c) * MC * 16 + at];
  rowsq[(int64_t)item * stride + l] = acc;
}

}  // namespace plsr
