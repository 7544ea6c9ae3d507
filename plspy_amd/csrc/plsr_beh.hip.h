// K4b: bootstrap of behaviour PLS in two stages, X in registers.
//
// K4a (plsr_agg.hip.h) multiplies X by the item's k x n operator  rows_b = (Yz_b U_c)^T.  In
// behaviour PLS that operator has rank <= b (behaviours) inside a cell c -- it is the
// reference's own two steps folded into one:
//
//     R_bc[beh, v] = sum_{i in c} Yz_b[i, beh] Z_b[i, v]          (class_functions.py:240-242)
//     VS_b[j, v]   = sum_c sum_beh U[(c, beh), j] R_bc[beh, v]    (bootstrap_permutation.py:620)
//
// With k = cells * b latent variables the folded product costs 2 k n p flops per item, the two
// steps 2 b n p + 2 k^2 p: at config 3 (k = 48, b = 8, n = 120) 0.38 + 0.92 instead of 2.3 GFLOP.
//
// Stage 1.  Yz_b is z-scored within cells, so its columns sum to zero over a cell and the shift of
// the z-score drops out: R_bc = sc_bc(v) * sum_r Yza_bc[beh, r] X'[r, v] with the aggregated Yza
// (rows of the sample summed per source row, as in K4a) -- an MFMA on the RAW centred X registers,
// no z-score FMA in the k-loop.  One 16-row MFMA tile holds the BP (8 or 16) behaviour rows of
// IP = 16 / BP items, which share the cell's k-steps.
// Stage 2.  The stage-1 accumulator of a cell, scaled by sc_bc(v), is -- register for register --
// the B operand of the projection on U: lane (voxel, g) holds rows g + 4 r of the tile, i.e.
// k-step r of that (item, cell) block.  So a cell's contribution VS += U_c^T R'_bc follows at once
// from the registers (BP / 4 k-steps x MC tiles per item), no transposition, no per-cell storage.
//
// X layout in registers: cell c owns fragments c * CSMAX ... (its source rows from src_lo[c] on,
// four per fragment, zero past src_hi[c]), so cells need not be aligned or equal; CSMAX * NCMAX <= 32.
// Statistics (four items per v_mfma_f64_4x4x4 on the same registers), staging of the per-group
// fragments through two LDS buffers, stores and moment sums as in K4a.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "plsr_agg.hip.h"

namespace plsr {

constexpr int BH_WAVES = 4;

struct BehMetaArgs {
  const double *Yz;                    // [items][nz][b]
  const double *U;                     // [ncell * b][k]
  const int32_t *src;                  // [items][nz]
  int32_t items, nz, b, k, ncell, cs, CSMAX, BP, IP, MC;
  int32_t cell_lo[AG_MAXZC + 1], src_lo[AG_MAXZC], src_hi[AG_MAXZC];
  double *a1;                          // [groups][ncell][cs][64]        stage-1 A fragments (Yz aggregated)
  double *mfrag;                       // [ceil(items / 4)][ncell][cs][64] multiplicities of four items
  double *u2;                          // [ncell][BP / 4][MC][64]        stage-2 A fragments (U^T)
};

__global__ __launch_bounds__(256) void beh_meta_kernel(BehMetaArgs A) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int ngrp = (A.items + A.IP - 1) / A.IP, nsg = (A.items + 3) / 4;
  const int64_t n1 = (int64_t)ngrp * A.ncell * A.cs * 64;
  const int64_t n2 = (int64_t)nsg * A.ncell * A.cs * 64;
  const int64_t n3 = (int64_t)A.ncell * (A.BP / 4) * A.MC * 64;
  if (e < n1) {
    const int lane = (int)(e & 63);
    int64_t t_ = e >> 6;
    const int t = (int)(t_ % A.cs);
    t_ /= A.cs;
    const int c = (int)(t_ % A.ncell);
    const int grp = (int)(t_ / A.ncell);
    const int m = lane & 15, kk = lane >> 4;
    const int item = grp * A.IP + m / A.BP, beh = m % A.BP;
    const int r = A.src_lo[c] + 4 * t + kk;
    double a = 0.0;
    if (item < A.items && beh < A.b && r < A.src_hi[c]) {
      const int32_t *s = A.src + (int64_t)item * A.nz;
      const double *y = A.Yz + (int64_t)item * A.nz * A.b + beh;
      for (int i = A.cell_lo[c]; i < A.cell_lo[c + 1]; ++i) a += s[i] == r ? y[(int64_t)i * A.b] : 0.0;
    }
    A.a1[e] = a;
  } else if (e - n1 < n2) {
    const int64_t q = e - n1;
    const int lane = (int)(q & 63);
    int64_t t_ = q >> 6;
    const int t = (int)(t_ % A.cs);
    t_ /= A.cs;
    const int c = (int)(t_ % A.ncell);
    const int sg = (int)(t_ / A.ncell);
    const int item = sg * 4 + (lane & 3);
    const int r = A.src_lo[c] + 4 * t + (lane >> 4);
    double m = 0.0;
    if (item < A.items && r < A.src_hi[c]) {
      const int32_t *s = A.src + (int64_t)item * A.nz;
      int cnt = 0;
      for (int i = A.cell_lo[c]; i < A.cell_lo[c + 1]; ++i) cnt += s[i] == r;
      m = (double)cnt;
    }
    A.mfrag[q] = m;
  } else if (e - n1 - n2 < n3) {
    const int64_t q = e - n1 - n2;
    const int lane = (int)(q & 63);
    int64_t t_ = q >> 6;
    const int mc = (int)(t_ % A.MC);
    t_ /= A.MC;
    const int qq = (int)(t_ % (A.BP / 4));
    const int c = (int)(t_ / (A.BP / 4));
    const int j = mc * 16 + (lane & 15);
    const int beh = 4 * qq + (lane >> 4);
    A.u2[q] = (beh < A.b && j < A.k) ? A.U[((int64_t)c * A.b + beh) * A.k + j] : 0.0;
  }
}

// a source row outside its cell's declared range: poison the item's group (see agg_check_kernel)
__global__ __launch_bounds__(256) void beh_check_kernel(BehMetaArgs A) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)A.items * A.nz) return;
  const int item = (int)(e / A.nz);
  const int i = (int)(e - (int64_t)item * A.nz);
  int c = 0;
  while (c + 1 < A.ncell && i >= A.cell_lo[c + 1]) ++c;
  const int r = A.src[e];
  if (r < A.src_lo[c] || r >= A.src_hi[c]) {
    // every row of the item's behaviour block in the group's first fragment
    const int m0 = (item % A.IP) * A.BP;
    for (int m = m0; m < m0 + A.BP; ++m) A.a1[(int64_t)(item / A.IP) * A.ncell * A.cs * 64 + m] = __builtin_nan("");
  }
}

struct BehArgs {
  const double *X;
  int64_t ldx, p;
  int32_t n, items, k, per;            // per = items per split (a multiple of 4)
  int32_t ncell, cs, MC;               // cells, k-steps per cell (<= CSMAX), tiles of latent variables
  int32_t src_lo[AG_MAXZC], src_hi[AG_MAXZC];
  double cnt[AG_MAXZC], rcnt[AG_MAXZC];
  const double *a1, *mfrag, *u2;
  double *S1, *S2;                     // [split][p][k] plain partial sums (overwritten), or null
  double *vst;                         // [items][k][ldv], tiled: [items][ldv / 32][k][32]; or null
  int64_t ldv;
  int32_t vst_tiled;                   // tile-major VS^T (K5i's operand layout; ldv a multiple of 32)
#ifdef BEH_TIMING
  long long *dbg;                      // developer-only: [workgroup][wave][8] cycle counts
#endif
};

// EXACT: the launch has exactly NCMAX cells of CSMAX k-steps and three tiles of latent variables (the
// BASELINE shape: six cells of twenty rows, k = 48), so every guard on the step counts is true at
// compile time -- the guards cost a compare and a branch beside each MFMA (3.5 SALU instructions per
// MFMA in the counters of the generic instance).  The basic-block boundaries the guards gave the
// scheduler are kept as scheduling barriers (without them it hoists the operand loads and spills).
template <int CSMAX, int NCMAX, int BP, bool EXACT = false>
__global__ __launch_bounds__(BH_WAVES * 64, 2) void item_beh_kernel(BehArgs A) {
  constexpr int NF = CSMAX * NCMAX;                // fragments of X in registers
  constexpr int IP = 16 / BP;                      // items per stage-1 tile
  constexpr int QB = BP / 4;                       // stage-2 k-steps per (item, cell)
  constexpr int MCM = 3;
  static_assert(NF <= 32 && (BP == 8 || BP == 16), "register budget / tile packing");
  typedef double d2 __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15;
  const int g = lane >> 4;
  const int64_t tile = (int64_t)blockIdx.x * BH_WAVES + wave;
  const int64_t v = tile * 16 + col;
  const bool vok = v < A.p;
  const int64_t vc = vok ? v : A.p - 1;
  const int it_lo = blockIdx.y * A.per;
  const int it_hi = min(A.items, it_lo + A.per);
  if (it_lo >= it_hi) return;
  const int ncell = EXACT ? NCMAX : A.ncell, cs = EXACT ? CSMAX : A.cs, MC = EXACT ? 3 : A.MC;
  const int unit = ncell * cs * 64;                // doubles of a group's stage-1 fragments
  const int unit_p = (unit + 511) / 512 * 512;     // ... padded to whole 16-byte pieces of 256 threads
  // LDS: [U2: ncell * QB * MC * 64][two stage-1 units][scale: waves x ncell x 64]
  double *u2s = smem;
  double *bufs = smem + (size_t)ncell * QB * MC * 64;
  double *st = bufs + 2 * unit_p + (size_t)wave * ncell * 64;    // [cell][item-in-four * 16 + voxel] scale
  // rows of the cells and their reciprocals: from LDS, not from the kernel arguments (sixteen cells'
  // worth of them held in SGPRs pushed the wide layouts into scratch)
  double *ctab = bufs + 2 * unit_p + (size_t)BH_WAVES * ncell * 64;
  if (tid < 2 * AG_MAXZC) ctab[tid] = (tid & 1) ? A.rcnt[tid >> 1] : A.cnt[tid >> 1];

  // ---- X in registers, cell by cell, centred by the voxel's grand mean over all n rows ----
  double x[NF];
  double d = 0.0;
  {
    // grand mean: every row once (a row may belong to no cell or to several)
    double acc = 0.0;
    for (int row = g; row < A.n; row += 4) acc += A.X[(int64_t)row * A.ldx + vc];
    acc += __shfl_xor(acc, 16);
    acc += __shfl_xor(acc, 32);
    d = acc / (double)A.n;
  }
#pragma unroll
  for (int c = 0; c < NCMAX; ++c)
#pragma unroll
    for (int t = 0; t < CSMAX; ++t) {
      const int row = (c < ncell ? A.src_lo[c] : 0) + 4 * t + g;
      const bool ok = c < ncell && t < cs && row < A.src_hi[c];
      const double xv = A.X[(int64_t)min(row, A.n - 1) * A.ldx + vc];
      x[c * CSMAX + t] = ok ? xv - d : 0.0;
    }

  // stage-2 fragments (U^T): resident in LDS for the life of the workgroup
  for (int e = tid; e < ncell * QB * MC * 64; e += BH_WAVES * 64) u2s[e] = A.u2[e];

  double s1[MCM][4], s2[MCM][4];            // plain moment sums over this split's items
  const bool moments = A.S1 != nullptr;
#pragma unroll
  for (int mc = 0; mc < MCM; ++mc)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s1[mc][r] = 0.0;
      s2[mc][r] = 0.0;
    }

  // ---- statistics of four items: scale of every cell -> LDS (one pass + exact redo, see K4a) ----
  auto stats_group = [&](int sg) {
    const double *mp = A.mfrag + ((size_t)sg * ncell * cs) * 64 + lane;
    uint32_t susp = 0;
    int ncell_ = ncell, cs_ = cs;
    if (!EXACT) asm volatile("" : "+s"(ncell_), "+s"(cs_));          // (see the group loop)
    // the multiplicity fragments of cell c + 1 are fetched while cell c is summed (an L2 round trip
    // per cell was the larger part of this phase)
    double mfn[CSMAX];
#pragma unroll
    for (int t = 0; t < CSMAX; ++t) mfn[t] = mp[(size_t)(t < cs_ ? t : 0) * 64];
#pragma unroll
    for (int c = 0; c < NCMAX; ++c) {
      if (c < ncell_) {
        double s1a = 0.0, s2a = 0.0, s1b = 0.0, s2b = 0.0;
#pragma unroll
        for (int t = 0; t < CSMAX; ++t) {
          const double m = mfn[t];
          mfn[t] = mp[(size_t)(cs_ + (t < cs_ ? t : 0)) * 64];     // slot t: the next cell's, right after its use
          if (t < cs_) {
            const double xv = x[c * CSMAX + t];
            if (t & 1) {
              s1b = mfma4_f64(m, xv, s1b);
              s2b = mfma4_f64(m, xv * xv, s2b);
            } else {
              s1a = mfma4_f64(m, xv, s1a);
              s2a = mfma4_f64(m, xv * xv, s2a);
            }
          }
          if (EXACT) __builtin_amdgcn_sched_barrier(0);
        }
        const double s1c = s1a + s1b, s2c = s2a + s2b;
        const double cnt = ctab[2 * c];
        const double mu = s1c * ctab[2 * c + 1];
        const double var = fma(-s1c, mu, s2c);
        const double em = 2.220446049250313e-16 * fabs(mu + d);
        const bool dead = !(var > cnt * em * em);
        if (s2c > 0.0 && !(var > 6.1e-5 * s2c)) susp |= 1u << c;
        st[c * 64 + lane] = dead ? 0.0 : rsqrt(var);
        mp += (size_t)cs_ * 64;
      }
    }
    if (__builtin_amdgcn_ballot_w64(susp != 0) == 0) return;
    // rare: exact second pass (a sample of nearly equal rows far from the grand mean)
    mp = A.mfrag + ((size_t)sg * ncell * cs) * 64 + lane;
#pragma unroll
    for (int c = 0; c < NCMAX; ++c) {
      if (c < ncell) {
        const uint64_t hit = __builtin_amdgcn_ballot_w64((susp >> c) & 1u);
        if (hit != 0) {
          const double cnt = ctab[2 * c];
          double a1 = 0.0;
#pragma unroll
          for (int t = 0; t < CSMAX; ++t)
            if (t < cs) a1 = mfma4_f64(mp[(size_t)t * 64], x[c * CSMAX + t], a1);
          const double mu4 = a1 / cnt;
          for (int i = 0; i < 4; ++i) {
            if (((hit >> (16 * i)) & 0xffffull) == 0) continue;
            const double mui = __shfl(mu4, i * 16 + col);
            double b1 = 0.0, b2 = 0.0;
#pragma unroll
            for (int t = 0; t < CSMAX; ++t)
              if (t < cs) {
                const double m = mp[(size_t)t * 64];
                const double tt = x[c * CSMAX + t] - mui;
                b1 = mfma4_f64(m, tt, b1);
                b2 = mfma4_f64(m, tt * tt, b2);
              }
            if (g == i && ((susp >> c) & 1u)) {
              const double dm = b1 / cnt;
              const double var = fma(-b1, dm, b2);
              const double em = 2.220446049250313e-16 * fabs(mui + dm + d);
              st[c * 64 + lane] = !(var > cnt * em * em) ? 0.0 : rsqrt(var);
            }
          }
        }
        mp += (size_t)cs * 64;
      }
    }
  };

  // ---- stage-1 fragments: a unit per group of IP items, two LDS buffers ----
  const int npc = unit_p / 512;                    // 16-byte pieces per thread and unit (<= 4: NF <= 32)
  const d2 *gsrc = (const d2 *)(A.a1 + (size_t)(it_lo / IP) * unit) + tid;
  {
    d2 *dst = (d2 *)bufs + tid;
    for (int q = 0; q < npc; ++q) dst[q * (BH_WAVES * 64)] = gsrc[q * (BH_WAVES * 64)];
  }
  __syncthreads();
  // eight start phases spread over about one group's duration (see K4a: workgroups that run in step
  // store their results in the same instants and queue on the memory system)
  {
    const int mfmas = ncell * cs + ncell * 4 * MC;            // per group
    const int ticks = min(127, max(1, mfmas * 2 * 2 / 8));    // x 64 cycles per MFMA / 64 cycles per tick, two waves per SIMD, ~2x for the rest
    for (int q = (int)((blockIdx.x + blockIdx.y) & 7) * ticks; q > 0; q -= 8) __builtin_amdgcn_s_sleep(8);
  }

  const bool odd = col & 1;
  const int jl = g + 8 * (int)odd;
  // row-major: row j of the item at j * ldv; tile-major: the 32-voxel tile's k rows of 256 bytes side by side (a
  // workgroup's 64 voxels are then two contiguous blocks of k x 256 bytes per item instead of k pieces of 512)
  const int64_t rstride = A.vst_tiled ? 256 : A.ldv * 8;
  const uint32_t loff = (uint32_t)(jl * rstride + (A.vst_tiled ? ((v - odd) >> 5) * A.k * 256 + ((v - odd) & 31) * 8
                                                                 : (v - odd) * 8));
  const int vcode = (v - odd) + 1 < A.p ? 2 : ((v - odd) < A.p ? 1 : 0);
  // (wave-uniform) every lane stores a pair of voxels of a row that exists: the stores need no per-lane guards
  const bool whole = __builtin_amdgcn_ballot_w64(vcode != 2) == 0 && A.k == 16 * MC;

  int par = 0;
#ifdef BEH_TIMING
  long long t_stats = 0, t_mm = 0, t_bar = 0, t_epi = 0;
  const long long t_begin = clock64();
#define BEH_T(var, t0) var += clock64() - t0
#else
#define BEH_T(var, t0)
#endif
  stats_group(it_lo >> 2);
  for (int item0 = it_lo; item0 < it_hi; item0 += IP) {
#ifdef BEH_TIMING
    const long long tm0 = clock64();
#endif
    // opaque copies of the step counts: compared inside the loop (s_cmp + s_cbranch_scc) instead of the
    // comparisons being hoisted out of it as one 64-bit mask per guard -- two dozen SGPR pairs that
    // spilled into VGPR lanes and from there into scratch
    int ncell_ = ncell, cs_ = cs, MC_ = MC;
    if (!EXACT) asm volatile("" : "+s"(ncell_), "+s"(cs_), "+s"(MC_));
    const double *cur = bufs + par * unit_p + lane;
    d2 *nxt = (d2 *)(bufs + (par ^ 1) * unit_p) + tid;
    gsrc += unit / 2;                              // next group's pieces (the stream is padded by one unit)

    f64x4 acc2[IP][MCM];
#pragma unroll
    for (int ii = 0; ii < IP; ++ii)
#pragma unroll
      for (int mc = 0; mc < MCM; ++mc) acc2[ii][mc] = (f64x4){0.0, 0.0, 0.0, 0.0};

    const double *a1p = cur;
    const double *u2p = u2s + lane;
    const double *stl = st + ((item0 - it_lo) & 3) * 16 + col;    // scale of item ii: + ii * 16
    // Operands are fetched in batches AHEAD of the branches that guard the MFMAs (the step counts are
    // run-time values): left inside them, every MFMA sat behind its own ds_read + s_waitcnt.  The
    // stage-1 fragments of cell c + 1 and the U fragments / scales of cell c are in flight while cell
    // c's stage-1 MFMAs run; loads past a cell's last k-step re-read its first (value unused).
    double fan[CSMAX];
#pragma unroll
    for (int t = 0; t < CSMAX; ++t) fan[t] = a1p[(size_t)(t < cs_ ? t : 0) * 64];
#pragma unroll
    for (int c = 0; c < NCMAX; ++c) {
      if (c < ncell_) {
        a1p += (size_t)cs_ * 64;                                  // -> the next cell's fragments
        // this cell's U fragments and scales: in flight during its stage-1 MFMAs
        double uf[QB][MCM], scl[IP];
#pragma unroll
        for (int qq = 0; qq < QB; ++qq)
#pragma unroll
          for (int mc = 0; mc < MCM; ++mc) uf[qq][mc] = u2p[((size_t)(c * QB + qq) * MC_ + (mc < MC_ ? mc : 0)) * 64];
#pragma unroll
        for (int ii = 0; ii < IP; ++ii) scl[ii] = stl[c * 64 + ii * 16];
        f64x4 r1 = (f64x4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int t = 0; t < CSMAX; ++t) {
          if (t < cs_) r1 = mfma_f64(fan[t], x[c * CSMAX + t], r1);
          if (EXACT) __builtin_amdgcn_sched_barrier(0);
          fan[t] = a1p[(size_t)(t < cs_ ? t : 0) * 64];           // slot t: refilled right after its use (past the last cell: unused)
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ii = (4 * r) / BP, qq = ((4 * r) % BP) / 4;
          const double bop = r1[r] * scl[ii];
#pragma unroll
          for (int mc = 0; mc < MCM; ++mc) {
            if (mc < MC_) acc2[ii][mc] = mfma_f64(uf[qq][mc], bop, acc2[ii][mc]);
            if (EXACT) __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      if (EXACT) __builtin_amdgcn_sched_barrier(0);
    }

    // The scales of the NEXT four items are computed here, ahead of this group's stores: a wave's
    // memory operations retire in order, and the multiplicity loads of the statistics, issued right
    // behind an item's 12 KB of stores, waited until HBM had taken them (a quarter of the kernel).
#ifdef BEH_TIMING
    const long long ts0 = clock64();
    t_mm += ts0 - tm0;
#endif
    // the next group's fragments are loaded after the statistics (the previous group's stores are a whole
    // group old by now) and written at once: held in registers for the life of the group they pushed
    // four of the register layouts into scratch
    if (((item0 + IP - it_lo) & 3) == 0 && item0 + IP < it_hi) stats_group((item0 + IP) >> 2);
    d2 park[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (q < npc) park[q] = gsrc[q * (BH_WAVES * 64)];
#ifdef BEH_TIMING
    const long long tb0 = clock64();
    t_stats += tb0 - ts0;
#endif
    // the next group's fragments into the other buffer; everybody is done reading this one
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (q < npc) nxt[q * (BH_WAVES * 64)] = park[q];
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    par ^= 1;
#ifdef BEH_TIMING
    const long long te0 = clock64();
    t_bar += te0 - tb0;
#endif

    // ---- group done: acc2[ii][mc][r] = VS of item item0 + ii, rows 16 mc + g + 4 r, voxel col ----
#pragma unroll
    for (int ii = 0; ii < IP; ++ii) {
      const int item = item0 + ii;
      if (item < it_hi) {
#if defined(PLSR_ABLATE) && (PLSR_ABLATE & 131072)
        const char *base = (const char *)A.vst + (int64_t)(item & 1) * A.k * A.ldv * 8;   // timing only: two items' worth of lines
#else
        const char *base = (const char *)A.vst + (int64_t)item * A.k * A.ldv * 8;
#endif
#pragma unroll
        for (int mc = 0; mc < MCM; ++mc) {
          if (mc < MC) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const double val = acc2[ii][mc][r];
              if (moments) {
                s1[mc][r] += val;
                s2[mc][r] = fma(val, val, s2[mc][r]);
              }
            }
            if (A.vst != nullptr) {
#pragma unroll
              for (int h = 0; h < 2; ++h) {
                // (opaque copies: the compiler turned `odd ? acc[h] : acc[2 + h]` into an extraction at a per-lane
                // index, a chain of three selects per register half where one does)
                double lo_ = acc2[ii][mc][h], hi_ = acc2[ii][mc][2 + h];
                asm volatile("" : "+v"(lo_), "+v"(hi_));
#if defined(PLSR_ABLATE) && (PLSR_ABLATE & 262144)
                const d2 pc = (d2){lo_, hi_};                           // timing only: no exchange
#else
                const double got = swap_pair_f64(odd ? lo_ : hi_);
                const d2 pc = odd ? (d2){got, hi_} : (d2){lo_, got};
#endif
                const int jrow = 16 * mc + 4 * h;
                char *dst = (char *)base + jrow * rstride + loff;
                if (whole) {
#if defined(PLSR_ABLATE) && (PLSR_ABLATE & 524288)
                  *(d2 *)dst = pc;
#else
                  __builtin_nontemporal_store(pc, (d2 *)dst);          // streamed: K5 reads it back from HBM anyway
#endif
                } else if (jl + jrow < A.k) {
                  if (vcode == 2) {
                    __builtin_nontemporal_store(pc, (d2 *)dst);
                  } else if (vcode == 1) {
                    *(double *)dst = pc.x;
                  }
                }
              }
            }
          }
        }
      }
    }
#ifdef BEH_TIMING
    t_epi += clock64() - te0;
#endif
  }

#ifdef BEH_TIMING
  if (lane == 0 && A.dbg != nullptr) {
    long long *o = A.dbg + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * BH_WAVES + wave) * 8;
    o[0] = clock64() - t_begin;
    o[1] = t_stats;
    o[2] = t_mm;
    o[3] = t_bar;
    o[4] = t_epi;
    o[5] = it_hi - it_lo;
  }
#endif
  if (moments) {
    double *o1 = A.S1 + (int64_t)blockIdx.y * A.p * A.k;
    double *o2 = A.S2 + (int64_t)blockIdx.y * A.p * A.k;
#pragma unroll
    for (int mc = 0; mc < MCM; ++mc)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = mc * 16 + g + 4 * r;
        if (mc < MC && j < A.k && vok) {
          o1[v * A.k + j] = s1[mc][r];
          o2[v * A.k + j] = s2[mc][r];
        }
      }
  }
}

}  // namespace plsr
