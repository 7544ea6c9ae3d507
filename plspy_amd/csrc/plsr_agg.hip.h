// K4a: bootstrap of behaviour / multiblock PLS on AGGREGATED operators, X in registers.
//
// Same contract as K4f (plsr_fused.hip.h):  VS_b = rows_b Z_b  with Z_b the item's
// gathered, per-cell z-scored copy of X (class_functions.py:185-247 on X[inds],
// bootstrap_permutation.py:547-620), never materialised.  K4f gathers the item's rows
// from an LDS tile through a per-item offset table and reads per-(item, cell, voxel)
// statistics that a separate kernel wrote to HBM.  Here the gather is folded into the
// OPERATOR instead: a bootstrap sample draws the rows of a cell from a fixed range of
// source rows (resample.py:132-160: subjects within their group, the same draw for every
// condition), so with
//
//     A_bc[j, r] = sum_{i in cell c, src_b(i) = r} rows_b[j, i]     (aggregated operator)
//     m_bc[r]    = #{i in cell c : src_b(i) = r}                    (multiplicities)
//
//     VS_b[j, v] = sum_c  sc_bc(v) * sum_r A_bc[j, r] X[r, v]  +  sh_bc(v) * sum_r A_bc[j, r]
//     mean_bc(v) = sum_r m_bc[r] X[r, v] / n_c,   n_c var_bc(v) = sum_r m_bc[r] (X[r, v] - mean)^2
//
// every k-step of every item reads the SAME rows of X.  One wave owns 16 voxels for all
// items of its split and keeps X[:, 16 voxels] in registers (n <= 128: 2 VGPRs per four
// rows) as the B operand of v_mfma_f64_16x16x4; the item's aggregated operator fragments
// stream from L2 (every wave of the chip reads the same stream at about the same time);
// one z-score FMA per k-step feeds all MC tiles of latent variables.  The statistics of
// FOUR items at a time come from v_mfma_f64_4x4x4_4b (A = multiplicities of four items,
// B = the X registers, unchanged, and their squares): a cell of 20 rows costs 10 small
// MFMAs per four items instead of a dependent two-pass VALU chain per item, and nothing
// goes through HBM.  X is held centred by the per-voxel grand mean d (z-scores are shift
// invariant, copied cells add d back), so that the one-pass variance s2 - s1^2 / n_c
// cancels only by (cell mean - grand mean)^2 / var.
//
// Step program.  Source rows are cut into NF fragments of four rows.  A SWEEP walks all
// NF fragments once; the cells of a sweep have disjoint fragment ranges (cells that share
// a fragment, or overlap, go to different sweeps: rb = one sweep; mb = the copied task
// block in one sweep, the z-scored behaviour cells in a second).  Operator fragments of
// rows outside a cell's range are zero.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "plsr_fused.hip.h"

#ifndef AGG_ABLATE
#define AGG_ABLATE 0   // developer-only timing ablations (wrong results when non-zero): 1 no statistics,
                       // 2 no staging of the next unit, 4 no barrier, 8 no z-score FMA, 16 no epilogue
#endif

namespace plsr {

constexpr int AG_MAXSWEEP = 4;
constexpr int AG_MAXZC = 16;       // z-scored cells (their scale / shift: 1 KiB of LDS per wave each)
constexpr int AG_MAXROWS = 128;    // source rows (32 fragments of X in 64 VGPRs)
constexpr int AG_WAVES = 4;        // waves per workgroup (16 voxels each)
constexpr int AG_RING = 2;         // operands are read this many k-steps ahead
constexpr int AG_SRING = 10;       // multiplicity fragments of the statistics are fetched this many steps ahead

struct AggProgram {
  int32_t nsweep, nzsweep, nzc, pad;
  uint32_t start[AG_MAXSWEEP];     // bit f: a cell of the sweep starts at fragment f
  uint32_t zstart[AG_MAXSWEEP];    // bit f: ... and it is z-scored (else copied)
  uint32_t zend[AG_MAXSWEEP];      // z-sweep s (in order): bit f: a z-scored cell ends with fragment f
  int32_t zsweep[AG_MAXSWEEP];     // z-sweep s -> sweep
  double cnt[AG_MAXZC];            // rows of the z-scored cells, in the order the sweeps meet them
  double rcnt[AG_MAXZC];           // 1 / cnt
  int8_t zc_zs[AG_MAXZC], zc_flo[AG_MAXZC], zc_fhi[AG_MAXZC];   // their z-sweep and first / last fragment
  int8_t rowcell[AG_MAXSWEEP][AG_MAXROWS];   // (sweep, source row) -> cell or -1
};

// ---------------------------------------------------------------------------
// Aggregation of the operator rows (and the multiplicities) into MFMA fragments.
struct AggMetaArgs {
  const double *rows;                  // [items][k][nz]
  const int32_t *src;                  // [items][nz]
  int32_t items, k, nz, MC, NF, ncell, unit, nups, us;   // unit: doubles per staged unit; nups units of us k-steps per sweep
  int32_t cell_lo[FZ_MAXCELL + 1];
  int32_t src_lo[FZ_MAXCELL], src_hi[FZ_MAXCELL], cell_z[FZ_MAXCELL];
  AggProgram prog;
  double *afrag;                       // [items][nsweep][nups][unit]  (unit >= us * MC * 64)
  double *mfrag;                       // [ngroups][nzsweep][NF][64]
  AggProgram *prog_out;                // the program, for item_agg_kernel (which reads it from memory:
                                       // as a by-value argument it cost that kernel fifty spilled SGPRs)
};

__global__ __launch_bounds__(256) void agg_meta_kernel(AggMetaArgs A) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t na = (int64_t)A.items * A.prog.nsweep * A.nups * A.unit;
  const int ngroups = (A.items + 3) / 4;
  const int64_t nm = (int64_t)ngroups * A.prog.nzsweep * A.NF * 64;
  if (e < (int64_t)(sizeof(AggProgram) / 4)) ((int32_t *)A.prog_out)[e] = ((const int32_t *)&A.prog)[e];
  if (e < na) {
    // A operand of v_mfma_f64_16x16x4: lane (m = lane & 15, k = lane >> 4); units of one
    // (item, sweep): [fragment][tile][lane], padded to `unit` doubles
    const int64_t un = e / A.unit;
    const int w = (int)(e - un * A.unit);
    const int uu = (int)(un % A.nups);                 // unit of the sweep: k-steps uu * us ...
    const int64_t sweep_id = un / A.nups;
    const int fs = (w >> 6) / A.MC;                    // k-step inside the unit
    const int f = uu * A.us + fs;
    if (fs >= A.us || f >= A.NF) {
      A.afrag[e] = 0.0;
      return;
    }
    const int lane = w & 63;
    const int mc = (w >> 6) % A.MC;
    const int sw = (int)(sweep_id % A.prog.nsweep);
    const int item = (int)(sweep_id / A.prog.nsweep);
    const int j = mc * 16 + (lane & 15);
    const int r = 4 * f + (lane >> 4);
    const int c = r < AG_MAXROWS ? A.prog.rowcell[sw][r] : -1;
    double a = 0.0;
    if (c >= 0 && j < A.k) {
      const int32_t *s = A.src + (int64_t)item * A.nz;
      const double *w = A.rows + ((int64_t)item * A.k + j) * A.nz;
      for (int i = A.cell_lo[c]; i < A.cell_lo[c + 1]; ++i) a += s[i] == r ? w[i] : 0.0;
    }
    A.afrag[e] = a;
  } else if (e - na < nm) {
    // A operand of v_mfma_f64_4x4x4_4b: lane (k = lane >> 4, block = (lane >> 2) & 3, i = lane & 3);
    // the same four items in every block (the blocks are four groups of four voxels)
    const int64_t q = e - na;
    const int lane = (int)(q & 63);
    int64_t t = q >> 6;
    const int f = (int)(t % A.NF);
    t /= A.NF;
    const int zs = (int)(t % A.prog.nzsweep);
    const int grp = (int)(t / A.prog.nzsweep);
    const int item = grp * 4 + (lane & 3);
    const int r = 4 * f + (lane >> 4);
    const int sw = A.prog.zsweep[zs];
    const int c = r < AG_MAXROWS ? A.prog.rowcell[sw][r] : -1;
    double m = 0.0;
    if (c >= 0 && item < A.items && A.cell_z[c]) {
      const int32_t *s = A.src + (int64_t)item * A.nz;
      int cntr = 0;
      for (int i = A.cell_lo[c]; i < A.cell_lo[c + 1]; ++i) cntr += s[i] == r;
      m = (double)cntr;
    }
    A.mfrag[q] = m;
  }
}

// A source row outside its cell's declared range would silently drop out of the
// aggregated operator: poison the item instead (NaN in its first fragment reaches every
// output of the item and the moment sums).
__global__ __launch_bounds__(256) void agg_check_kernel(AggMetaArgs A) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)A.items * A.nz) return;
  const int item = (int)(e / A.nz);
  const int i = (int)(e - (int64_t)item * A.nz);
  int c = 0;
  while (c + 1 < A.ncell && i >= A.cell_lo[c + 1]) ++c;
  const int r = A.src[e];
  if (r < A.src_lo[c] || r >= A.src_hi[c])
    A.afrag[(int64_t)item * A.prog.nsweep * A.nups * A.unit] = __builtin_nan("");
}

// ---------------------------------------------------------------------------
struct AggArgs {
  const double *X;
  int64_t ldx, p;
  int32_t n, items, k, per;            // per = items per split (a multiple of 4)
  const AggProgram *__restrict__ prog; // (device memory, written by agg_meta_kernel)
  const double *afrag, *mfrag;
  double *S1, *S2;                     // [split][p][k] partial sums of VS, VS^2 (overwritten), or null
  double *vst;                         // [items][k][ldv] VS^T, or null
  int64_t ldv;
  double *rowsq_part;                  // [16-voxel tile][items][MC*16] or null
#ifdef AGG_TIMING
  long long *dbg;                      // developer-only: [workgroup][wave][8] cycle counts
#endif
};

// value of the neighbouring lane (lane ^ 1), by DPP quad_perm [1, 0, 3, 2] on the two halves
__device__ __forceinline__ double swap_pair_f64(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double mfma4_f64(double a, double b, double c) {
  return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}

// A UNIT is what the workgroup stages at a time: the operator fragments of US consecutive
// k-steps of one sweep of one item.  Two units and the scale / shift tables must leave room
// for a second workgroup on the CU (two independent workgroups are what overlaps one's
// statistics, epilogue and barrier waits with the other's MFMAs), so a sweep is cut into
// units of at most 45 (fragment, tile) pairs (22.5 KiB).
constexpr int agg_units_per_sweep(int nf, int mc) { return (nf * mc + 44) / 45; }
constexpr int agg_unit_steps(int nf, int mc) { return (nf + agg_units_per_sweep(nf, mc) - 1) / agg_units_per_sweep(nf, mc); }
// doubles of one staged unit, padded so that the workgroup's 256 threads move it in whole 16-byte pieces
constexpr int agg_unit_elems(int nf, int mc) { return (agg_unit_steps(nf, mc) * mc * 64 + 511) / 512 * 512; }

// Workgroup = AG_WAVES (4) waves = 64 voxels; every wave owns 16 voxels and all MC tiles of
// latent variables.  The operator fragments of a unit are the same for every wave: the
// workgroup copies unit u + 1 into the other of two LDS buffers while it computes unit u (each
// thread parks 16-byte pieces in registers for WDELAY k-steps; NQ pieces per unit) and meets at
// ONE barrier per unit.  A operands come from LDS with immediate offsets, two k-steps ahead;
// B = the wave's X registers through one z-score FMA per k-step, formed a step ahead.
template <int NF, int MC>
__global__ __launch_bounds__(AG_WAVES * 64, 2) void item_agg_kernel(AggArgs A) {
  constexpr int NUPS = agg_units_per_sweep(NF, MC);   // units per sweep
  constexpr int US = agg_unit_steps(NF, MC);          // k-steps per unit (the last unit of a sweep may be shorter)
  constexpr int UNIT = agg_unit_elems(NF, MC);
  constexpr int NQ = UNIT / 512;                   // 16-byte pieces per thread and unit
  constexpr int WDELAY = 2;                        // k-steps between a piece's load and its LDS write
  constexpr int NP = 2 * MC;                       // 16-byte pieces of an item's results per lane
  // Schedule inside a unit (k-step s of the unit).  A wave's memory operations retire in order and
  // a store to HBM takes some 5 000 cycles to be acknowledged here, so a load issued within eight
  // k-steps after a store is not seen to have landed until the store has (measured: 3 900 cycles
  // per item lost with the stores ahead of the loads).  Hence: the first unit of a sweep stages
  // its successor at once (loads at steps 0 .. NQ - 1, each written WDELAY steps later), THEN the
  // previous item's results are stored (steps S0 .. S0 + NP - 1: younger than every load wait of
  // the unit, which the compiler can therefore count exactly); the other units stage as late as
  // they can (from step SLB), eight or more k-steps behind those stores.
  constexpr int USL = NF - (NUPS - 1) * US;        // steps of the last (shortest) unit
  constexpr int SLB = USL - WDELAY - NQ > 0 ? USL - WDELAY - NQ : 0;
  constexpr int S0 = NQ + WDELAY < US - NP ? NQ + WDELAY : US - NP;
  static_assert(NQ - 1 + WDELAY < USL, "a unit's pieces are all staged within the previous unit's k-steps");
  static_assert(NP < US, "an item's results leave during the first k-steps of the next");
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15;
  const int g = lane >> 4;
  const int64_t tile = (int64_t)blockIdx.x * AG_WAVES + wave;     // 16-voxel tile
  const int64_t v = tile * 16 + col;
  const bool vok = v < A.p;
  const int64_t vc = vok ? v : A.p - 1;
  const int it_lo = blockIdx.y * A.per;
  const int it_hi = min(A.items, it_lo + A.per);
  if (it_lo >= it_hi) return;                      // (the whole workgroup)
  const AggProgram *__restrict__ P = A.prog;
  const int nsweep = P->nsweep, nzsweep = P->nzsweep, nzc = P->nzc;
  double *st = smem + 2 * UNIT + (size_t)wave * nzc * 128;   // [zc][scale, shift][item-in-group * 16 + voxel]
  double *ctab = smem + 2 * UNIT + (size_t)AG_WAVES * nzc * 128;   // [zc][rows, 1 / rows] of the z-scored cells
  if (tid < 2 * AG_MAXZC) ctab[tid] = (tid & 1) ? P->rcnt[tid >> 1] : P->cnt[tid >> 1];

  // ---- X[:, 16 voxels] as B operand fragments: lane (n = voxel col, k = row g of the fragment) ----
  double x[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    const int row = 4 * f + g;
    x[f] = A.X[(int64_t)min(row, A.n - 1) * A.ldx + vc];
    if (row >= A.n) x[f] = 0.0;
  }
  // centred by the grand mean of the voxel
  double d = 0.0;
#pragma unroll
  for (int f = 0; f < NF; ++f) d += x[f];
  d += __shfl_xor(d, 16);
  d += __shfl_xor(d, 32);
  d /= (double)A.n;
#pragma unroll
  for (int f = 0; f < NF; ++f) x[f] = 4 * f + g < A.n ? x[f] - d : 0.0;

  // plain sums over this split's items; the shift by the observed VS is applied when the
  // partials are merged (moment_unshift_kernel)
  double s1[MC][4], s2[MC][4];
  const bool moments = A.S1 != nullptr;
#pragma unroll
  for (int mc = 0; mc < MC; ++mc)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s1[mc][r] = 0.0;
      s2[mc][r] = 0.0;
    }

  // ---- statistics of four items: scale / shift of every z-scored cell -> LDS ----
  auto stats_group = [&](int grp) {
    const double *mp = A.mfrag + ((size_t)grp * nzsweep * NF) * 64 + lane;
    const double *mp0 = mp;
    uint32_t susp = 0;              // bit zc: this lane's one-pass variance of cell zc cancelled too far
    int zc = 0;
    double cnt = ctab[0], rcnt = ctab[1];            // of the cell being summed (fetched a cell ahead)
    for (int zs = 0; zs < nzsweep; ++zs) {
      const uint32_t endm = P->zend[zs];
      // (a step is two small MFMAs: the multiplicities are fetched AG_SRING steps ahead to cover an
      // L2 round trip; the accumulators of the main loop are not live here, so the registers exist)
      double mf[AG_SRING];
#pragma unroll
      for (int u = 0; u < AG_SRING; ++u) mf[u] = mp[(size_t)u * 64];
      // (two accumulation chains per sum: a chain of dependent 4x4x4 MFMAs exposes its latency every time)
      double s1a = 0.0, s1b = 0.0, s2a = 0.0, s2b = 0.0;
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const double m = mf[f % AG_SRING];
        mf[f % AG_SRING] = mp[(size_t)(f + AG_SRING) * 64];        // (the stream carries AG_SRING steps of padding)
        if (f & 1) {
          s1b = mfma4_f64(m, x[f], s1b);
          s2b = mfma4_f64(m, x[f] * x[f], s2b);
        } else {
          s1a = mfma4_f64(m, x[f], s1a);
          s2a = mfma4_f64(m, x[f] * x[f], s2a);
        }
        if ((endm >> f) & 1u) {
          // lane (item g of the group, voxel col): s1c = sum m x', s2c = sum m x'^2 over the cell
          const double s1c = s1a + s1b, s2c = s2a + s2b;
          const double mu = s1c * rcnt;
          const double var = fma(-s1c, mu, s2c);                 // n_c * variance
          // scipy.stats.zscore's constant-slice rule (sd <= eps |mean|) followed by nan_to_num -> 0
          const double em = 2.220446049250313e-16 * fabs(mu + d);
          const bool dead = !(var > cnt * em * em);
          // the one-pass variance is good to 4 eps s2c / var: a sample of (nearly) equal rows far
          // from the grand mean -- duplicates in a small cell -- is redone below with its own shift
          if (s2c > 0.0 && !(var > 6.1e-5 * s2c)) susp |= 1u << zc;
          const double sc = dead ? 0.0 : rsqrt(var);
          st[(zc * 2) * 64 + lane] = sc;
          st[(zc * 2 + 1) * 64 + lane] = dead ? 0.0 : -mu * sc;
          s1a = s1b = s2a = s2b = 0.0;
          zc = min(zc + 1, AG_MAXZC - 1);
          cnt = ctab[2 * zc];
          rcnt = ctab[2 * zc + 1];
        }
      }
      mp += (size_t)NF * 64;
    }
    if (__builtin_amdgcn_ballot_w64(susp != 0) == 0) return;
    // ---- rare: exact second pass for the flagged (item, cell, voxel) triples ----
    for (zc = 0; zc < nzc; ++zc) {
      const uint64_t hit = __builtin_amdgcn_ballot_w64((susp >> zc) & 1u);
      if (hit == 0) continue;
      const int flo = P->zc_flo[zc], fhi = P->zc_fhi[zc];
      const double cnt = P->cnt[zc];
      const double *mq = mp0 + (size_t)P->zc_zs[zc] * NF * 64;
      double a1 = 0.0;
#pragma unroll
      for (int f = 0; f < NF; ++f)
        if (f >= flo && f <= fhi) a1 = mfma4_f64(mq[(size_t)f * 64], x[f], a1);
      const double mu4 = a1 / cnt;                   // lane (item g, voxel col): mean about the grand mean
      for (int i = 0; i < 4; ++i) {
        if (((hit >> (16 * i)) & 0xffffull) == 0) continue;
        const double mui = __shfl(mu4, i * 16 + col);   // item i's mean at this lane's voxel, in all four row lanes
        double b1 = 0.0, b2 = 0.0;
#pragma unroll
        for (int f = 0; f < NF; ++f)
          if (f >= flo && f <= fhi) {
            const double m = mq[(size_t)f * 64];
            const double t = x[f] - mui;
            b1 = mfma4_f64(m, t, b1);
            b2 = mfma4_f64(m, t * t, b2);
          }
        if (g == i && ((susp >> zc) & 1u)) {
          const double dm = b1 / cnt;
          const double var = fma(-b1, dm, b2);
          const double mu = mui + dm;
          const double em = 2.220446049250313e-16 * fabs(mu + d);
          const bool dead = !(var > cnt * em * em);
          const double sc = dead ? 0.0 : rsqrt(var);
          st[(zc * 2) * 64 + lane] = sc;
          st[(zc * 2 + 1) * 64 + lane] = dead ? 0.0 : -mu * sc;
        }
      }
    }
  };

  // ---- operator fragments: units staged through two LDS buffers ----
  typedef double d2 __attribute__((ext_vector_type(2)));
  const d2 *gsrc = (const d2 *)(A.afrag + (size_t)it_lo * nsweep * NUPS * UNIT) + tid;   // this thread's pieces of unit 0
  {
    d2 *dst = (d2 *)smem + tid;
#pragma unroll
    for (int q = 0; q < NQ; ++q) dst[q * (AG_WAVES * 64)] = gsrc[q * (AG_WAVES * 64)];
  }
  __syncthreads();

  // Workgroups run the same schedule and start together: without an offset the whole chip would
  // store its results in the same few k-steps of every item (77 MB per item at config 3, a burst
  // several times the rate HBM takes).  Eight phases, an eighth of an item's MFMA time apart.
  for (int ph = (int)((blockIdx.x + blockIdx.y) & 7); ph > 0; --ph)
    __builtin_amdgcn_s_sleep((NF * MC * 64 * 2 / 8 + 63) / 64 > 127 ? 127 : (NF * MC * 64 * 2 / 8 + 63) / 64);

  // ---- results of an item leave the wave during the NEXT item's first k-steps ----
  // 8-byte stores are issue-bound (about 7 B / cycle / CU).  Two lanes that hold neighbouring
  // voxels exchange half of their values (even lane: rows r = 0, 1 of both voxels, odd lane:
  // rows 2, 3) and the item's VS goes out as 2 MC 16-byte stores per lane.
  const bool odd = col & 1;
  // address of a piece = uniform base of (item, 4-row group) + this lane's constant byte offset
  // (32 bits: the library checks (12 ldv + p) * 8 < 2^32), so that the store takes its base from
  // scalar registers and the k-loop carries no 64-bit vector address arithmetic
  const int jl = g + 8 * (int)odd;                 // this lane's row within the 16-row tile, for piece h: + 4 h
  const uint32_t loff = (uint32_t)(((int64_t)jl * A.ldv + (v - odd)) * 8);
  const int vcode = (v - odd) + 1 < A.p ? 2 : ((v - odd) < A.p ? 1 : 0);   // both voxels of the pair exist / the first only
  double hold[MC][4];                              // the previous item's VS (acc as it stood)
  int hold_item = -1;
  const char *hold_base = nullptr;                 // (uniform) &vst[hold_item][0][0]
  auto store_piece = [&](int mc, int h) {
    // even lane: row r = h of its own and its neighbour's voxel; odd lane: row r = 2 + h
    const double got = swap_pair_f64(odd ? hold[mc][h] : hold[mc][2 + h]);
    const d2 pc = odd ? (d2){got, hold[mc][2 + h]} : (d2){hold[mc][h], got};
    const int jrow = 16 * mc + 4 * h;
    if (jl + jrow < A.k) {
      char *dst = (char *)hold_base + (int64_t)jrow * A.ldv * 8 + loff;
      if (vcode == 2) {
        __builtin_nontemporal_store(pc, (d2 *)dst);            // streamed: K5 reads it back from HBM anyway
      } else if (vcode == 1) {
        *(double *)dst = pc.x;
      }
    }
  };
  // the moment sums take the previous item's values one (tile, row) pair per k-step
  constexpr int MPS = (4 * MC + US - 2) / (US - 1);    // pairs per k-step, from step 1
  auto moment_pairs = [&](int s_) {
#pragma unroll
    for (int u = 0; u < MPS; ++u) {
      const int idx = (s_ - 1) * MPS + u;
      if (idx >= 0 && idx < 4 * MC) {
        const double val = hold[idx >> 2][idx & 3];
        s1[idx >> 2][idx & 3] += val;
        if (!(AGG_ABLATE & 16)) s2[idx >> 2][idx & 3] = fma(val, val, s2[idx >> 2][idx & 3]);
      }
    }
  };

  int par = 0;                                     // buffer of the current unit
#ifdef AGG_TIMING
  long long t_stats = 0, t_loop = 0, t_bar = 0, t_epi = 0;
  const long long t_begin = clock64();
#define AGG_T(var, code) { const long long t0_ = clock64(); code; var += clock64() - t0_; }
#else
#define AGG_T(var, code) { code; }
#endif
  for (int item = it_lo; item < it_hi; ++item) {
    const int ig = (item - it_lo) & 3;
    AGG_T(t_stats, if (!(AGG_ABLATE & 1) && ig == 0 && nzc > 0) stats_group(item >> 2));          // it_lo is a multiple of 4
#ifdef AGG_TIMING
    const long long tl0 = clock64();
#endif
    f64x4 acc[MC];
#pragma unroll
    for (int mc = 0; mc < MC; ++mc) acc[mc] = (f64x4){0.0, 0.0, 0.0, 0.0};
    const double *stl = st + ig * 16 + col;
    // scale / shift of the next z-scored cell the sweeps will meet, fetched one cell ahead
    int zc = 0;
    double scn = 0.0, shn = 0.0;
    if (nzc > 0) {
      scn = stl[0];
      shn = stl[64];
    }
    for (int sw = 0; sw < nsweep; ++sw) {
      const uint32_t stm = P->start[sw], zm = P->zstart[sw];
      double sc = 0.0, sh = 0.0;
      // the B operand of a k-step is formed during the step before it (z-score FMA, and at a
      // cell's first fragment the switch to its scale / shift), behind that step's MFMAs
      auto enter = [&](int f) {
        if ((stm >> f) & 1u) {                  // (wave-uniform)
          if ((zm >> f) & 1u) {
            sc = scn;
            sh = shn;
            zc = min(zc + 1, nzc - 1);
            scn = stl[(zc * 2) * 64];
            shn = stl[(zc * 2 + 1) * 64];
          } else {
            sc = 1.0;
            sh = d;
          }
        }
        return (AGG_ABLATE & 8) ? x[f] : fma(x[f], sc, sh);
      };
      double z = enter(0);
      const double *cur = smem + par * UNIT + lane;               // A fragments of the current unit
      d2 *nxt = (d2 *)(smem + (par ^ 1) * UNIT) + tid;            // where the next unit goes
      gsrc += UNIT / 2;                                           // the next unit's pieces (the stream is padded by one unit)
      double fa[AG_RING][MC];
#pragma unroll
      for (int u = 0; u < AG_RING; ++u)
#pragma unroll
        for (int mc = 0; mc < MC; ++mc) fa[u][mc] = cur[(u * MC + mc) * 64];
      d2 park[WDELAY];
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const int un = f / US, s_ = f - un * US;                  // unit of the sweep, k-step inside it
        const int ulen = un == NUPS - 1 ? USL : US;
        // The non-MFMA work of a k-step sits in the shadows of the step's own MFMAs (an MFMA holds
        // the matrix pipe for 64 cycles, the wave is free to issue other instructions meanwhile).
        // Left to the scheduler, a step was MC MFMAs followed by everything else, the two waves
        // of a SIMD fell into step with each other and the pipe idled through both waves'
        // "everything else" (about 120 of 500 cycles per k-step).
        double znext = 0.0;
#pragma unroll
        for (int mc = 0; mc < MC; ++mc) {
          acc[mc] = mfma_f64(fa[s_ % AG_RING][mc], z, acc[mc]);
          __builtin_amdgcn_sched_barrier(0);
          if (s_ + AG_RING < ulen) fa[s_ % AG_RING][mc] = cur[((s_ + AG_RING) * MC + mc) * 64];
          if (mc == 0 && !(AGG_ABLATE & 2)) {
            // staging of the next unit: piece q is loaded at step sl0 + q and written WDELAY steps later
            const int sl0 = un == 0 ? 0 : SLB;
            const int qw = s_ - sl0 - WDELAY, ql = s_ - sl0;
            if (qw >= 0 && qw < NQ) nxt[qw * (AG_WAVES * 64)] = park[qw % WDELAY];
            if (ql >= 0 && ql < NQ) park[ql % WDELAY] = gsrc[ql * (AG_WAVES * 64)];
          }
          if (mc == (MC > 1 ? 1 : 0) && f + 1 < NF) znext = enter(f + 1);
          if (mc == MC - 1 && un == 0) {
            // the previous item's results (first unit of the item only): moment sums from step 1,
            // stores at k-steps S0 .. S0 + NP - 1
            if (s_ >= 1 && (s_ - 1) * MPS < 4 * MC) {
              if (sw == 0 && hold_item >= 0 && moments) moment_pairs(s_);
            }
            if (!(AGG_ABLATE & 16) && s_ >= S0 && s_ < S0 + NP) {
              if (sw == 0 && hold_item >= 0 && A.vst != nullptr) store_piece((s_ - S0) >> 1, (s_ - S0) & 1);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        z = znext;
        if (s_ == ulen - 1) {
          // unit done: everybody has read it and has written its pieces of the next (LDS only:
          // global stores stay in flight across the barrier)
#ifdef AGG_TIMING
          const long long tb0 = clock64();
#endif
          if (!(AGG_ABLATE & 4)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef AGG_TIMING
          t_bar += clock64() - tb0;
#endif
          par ^= 1;
          if (un + 1 < NUPS) {
            cur = smem + par * UNIT + lane;
            nxt = (d2 *)(smem + (par ^ 1) * UNIT) + tid;
            gsrc += UNIT / 2;
#pragma unroll
            for (int u = 0; u < AG_RING; ++u)
#pragma unroll
              for (int mc = 0; mc < MC; ++mc) fa[u][mc] = cur[(u * MC + mc) * 64];
          }
        }
      }
    }
#ifdef AGG_TIMING
    const long long te0 = clock64();
    t_loop += te0 - tl0;
#endif

    // ---- item done: acc[mc][r] = VS[j = 16 mc + g + 4 r][voxel col]; it is consumed (moment
    // sums, stores) during the next item's first k-steps ----
#pragma unroll
    for (int mc = 0; mc < MC; ++mc) {
      if (A.rowsq_part != nullptr) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          double t = vok ? acc[mc][r] * acc[mc][r] : 0.0;
          t += __shfl_xor(t, 1);
          t += __shfl_xor(t, 2);
          t += __shfl_xor(t, 4);
          t += __shfl_xor(t, 8);
          if (col == 0) A.rowsq_part[(tile * A.items + item) * (MC * 16) + mc * 16 + g + 4 * r] = t;
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) hold[mc][r] = acc[mc][r];
    }
    hold_item = item;
    hold_base = (const char *)A.vst + (int64_t)item * A.k * A.ldv * 8;
#ifdef AGG_TIMING
    t_epi += clock64() - te0;
#endif
  }
#ifdef AGG_TIMING
  if (lane == 0 && A.dbg != nullptr) {
    long long *o = A.dbg + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * AG_WAVES + wave) * 8;
    o[0] = clock64() - t_begin;
    o[1] = t_stats;
    o[2] = t_loop;
    o[3] = t_bar;
    o[4] = t_epi;
    o[5] = it_hi - it_lo;
    o[6] = t_begin;
  }
#endif
  // the last item's values
  if (moments) {
#pragma unroll
    for (int s_ = 1; (s_ - 1) * MPS < 4 * MC; ++s_) moment_pairs(s_);
  }
  if (A.vst != nullptr && !(AGG_ABLATE & 16)) {
#pragma unroll
    for (int mc = 0; mc < MC; ++mc) {
      store_piece(mc, 0);
      store_piece(mc, 1);
    }
  }

  if (moments) {
    double *o1 = A.S1 + (int64_t)blockIdx.y * A.p * A.k;
    double *o2 = A.S2 + (int64_t)blockIdx.y * A.p * A.k;
#pragma unroll
    for (int mc = 0; mc < MC; ++mc)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = mc * 16 + g + 4 * r;
        if (j < A.k && vok) {
          o1[v * A.k + j] = s1[mc][r];
          o2[v * A.k + j] = s2[mc][r];
        }
      }
  }
}

}  // namespace plsr
