// K2: per-item cross-block Grams and their eigen-decomposition (thin SVD).
//
// An *item* is one decomposition request: m operator rows A (m x n) whose
// cross-block is M = A X (m x p).  The kernel returns G = M M^T (m x m) without
// ever storing M:
//
//   first product   D[v, c] = sum_i X[i, v] A[c, i]     MFMA: M = 16 voxels,
//                                                        N = 16 rows of A, K = 4
//   Gram            G[c1, c2] += sum_v D[v, c1] D[v, c2]
//
// With the f64 16x16x4 accumulator map D[row = (lane>>4)+4*reg][col = lane&15]
// the four registers of the first product are, unchanged, both the A and the B
// operand of the Gram MFMA for k-step `reg` (cdna guide section 3, "an
// accumulator tile as the next MFMA's operand"), so the Gram needs no LDS
// round trip.  For a split-half item the rows are the two halves' operators
// stacked, and G holds M1 M1^T, M1 M2^T and M2 M2^T at once.
//
// Reference arithmetic replaced: class_functions.py:98-123 (_run_pls ->
// np.linalg.svd) as used by split_half_resampling.py:194-196, :612-613,
// :682-683 and by the PLS constructors (pls_classes.py:261).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "plsr_project.hip.h"

namespace plsr {

struct GramArgs {
  const double *X;
  int64_t x_item_stride;     // 0: every item uses the same X; else doubles between items' matrices
  int64_t ldx, p;
  int32_t n, nk;
  const double *frag;    // [items*MC][nk][64], rows layout
  int32_t items;
  int32_t tiles_per_chunk;   // 64-voxel tiles per voxel chunk
  int32_t ks;                // k-steps of X staged in LDS at a time (rows of X are K-chunked)
  double *G_part;        // [nchunk][items][mm][mm]
  // fused gather / z-score (template FUSED): the item's matrix is never stored;
  // row r of item b is X[src[b][r]] * sc[b][cell(r)] + sh[b][cell(r)]  (plsr_fused.hip.h)
  const int32_t *src;        // [items][n]
  const int32_t *rowcell;    // [n] cell of every row
  int32_t ncell;
  const double *sc, *sh;     // [items][ncell][p]
  // block-sparse operators (template SL / SH): bit s of act[0] -- some item's operator is non-zero in
  // k-step s of a tile below SL; act[1] -- of a tile from SH on (gram_activity_kernel)
  const uint64_t *act;
  // FUSED: per (item, row) the addresses of the row's source row in X and of its cell's rows in sc / sh
  // (gram_rowtab_kernel): [items][4 nk + GRAM_PF] entries of four words, read through the scalar cache
  const int64_t *rowtab;
};

constexpr int GRAM_PF = 20;   // rows of X a thread parks in registers per K-chunk (ks <= GRAM_PF)

// rowtab[item][row] = { &X[src[item][row]][0], &sc[item][cell(row)][0], &sh[item][cell(row)][0], 0 } for row <
// 4 nk + GRAM_PF, rows from n on repeating row n - 1: a wave reads entries rbeg .. rbeg + PF - 1 without a clamp
__global__ __launch_bounds__(256) void gram_rowtab_kernel(const int32_t *src, const int32_t *rowcell, int items, int n,
                                                          int ntr, int ncell, const double *X, int64_t ldx,
                                                          const double *sc, const double *sh, int64_t p, int64_t *tab) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)items * ntr) return;
  const int item = (int)(e / ntr), row = min((int)(e % ntr), n - 1);
  const int64_t cell_row = ((int64_t)item * ncell + rowcell[row]) * p;
  tab[4 * e] = (int64_t)(uintptr_t)(X + (int64_t)src[(int64_t)item * n + row] * ldx);
  tab[4 * e + 1] = (int64_t)(uintptr_t)(sc + cell_row);
  tab[4 * e + 2] = (int64_t)(uintptr_t)(sh + cell_row);
  tab[4 * e + 3] = 0;
}

// ... except for six-tile items that stage their own gathered / z-scored rows (m = 81..96, e.g.
// behaviour split-half with k = 48): 21 Gram tiles + three parked values per row spilled
// 124 bytes per lane at 20 rows (16 at 10); 8 rows fit
constexpr int gram_pf(int mc, bool own_rows) { return (own_rows && mc >= 6) ? 8 : GRAM_PF; }

// Which k-steps of which tile groups carry anything: the operator rows of a split-half item are two
// halves that touch disjoint source rows (split_half_resampling._items_rb / _items_mb), so half of the
// first product's MFMAs would multiply zeros.  Tiles [0, SL) form the LOW group, [SL, SH) are always
// computed, [SH, MC) form the HIGH group; a k-step skips a group none of whose operator fragments (over
// all items of the launch) holds a non-zero.  Found from the fragments themselves, so any operator is
// handled correctly -- a dense one just gets all-ones masks.
__global__ __launch_bounds__(256) void gram_activity_kernel(const double *frag, int64_t total, int MC, int nk, int SL,
                                                            int SH, unsigned long long *act) {
  unsigned long long lo = 0, hi = 0;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    if (frag[e] != 0.0) {
      const int64_t ts = e >> 6;                          // (item * MC + tile) * nk + step
      const int step = (int)(ts % nk), tile = (int)((ts / nk) % MC);
      if (tile < SL) lo |= 1ull << step;
      if (tile >= SH) hi |= 1ull << step;
    }
  }
  for (int sft = 32; sft > 0; sft >>= 1) {
    lo |= __shfl_xor(lo, sft);
    hi |= __shfl_xor(hi, sft);
  }
  if ((threadIdx.x & 63) == 0) {
    if (lo) atomicOr(act, lo);
    if (hi) atomicOr(act + 1, hi);
  }
}

// MC = 16-row tiles per item, B = items per workgroup; SL / SH: see gram_activity_kernel (0 / MC: dense)
template <int MC, int B, bool FUSED = false, int SL = 0, int SH = MC>
__global__ __launch_bounds__(256, 1) void gram_kernel(GramArgs A) {
  static_assert(!FUSED || B == 1, "fused items do not share a tile");
  static_assert(0 <= SL && SL <= SH && SH <= MC, "tile groups");
  constexpr bool SPARSE = SL > 0 || SH < MC;
  unsigned long long act_lo = ~0ull, act_hi = ~0ull;
  if (SPARSE) {
    act_lo = A.act[0];
    act_hi = A.act[1];
  }
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int NG = MC * (MC + 1) / 2;
  constexpr int MM = MC * 16;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15;
  const int g = lane >> 4;
  const int nrows = A.nk * 4;
  const int item0 = blockIdx.x * B;
  const double *Xi = A.X + (int64_t)item0 * A.x_item_stride;   // B == 1 whenever the stride is non-zero
  const int chunk = blockIdx.y;

  double *ops = smem;                                   // [B][MC][nk][64]
  double *Xs = smem + (size_t)B * MC * A.nk * 64;       // [4*ks][64]

  // operator fragments of this workgroup's items -> LDS (once)
  {
    const int64_t total = (int64_t)B * MC * A.nk * 64;
    for (int64_t e = tid; e < total; e += 256) {
      const int b = (int)(e / ((int64_t)MC * A.nk * 64));
      const int item = item0 + b;
      ops[e] = item < A.items ? A.frag[(int64_t)item0 * MC * A.nk * 64 + e] : 0.0;
    }
  }

  f64x4 G[B][NG];
#pragma unroll
  for (int b = 0; b < B; ++b)
#pragma unroll
    for (int i = 0; i < NG; ++i) G[b][i] = (f64x4){0.0, 0.0, 0.0, 0.0};

  const int64_t nvt = (A.p + TV - 1) / TV;
  const int64_t t_lo = (int64_t)chunk * A.tiles_per_chunk;
  const int64_t t_hi = min(nvt, t_lo + A.tiles_per_chunk);
  const int xo = xs_index(g, wave * 16 + col);          // A operand: voxel = col of this wave's 16, row = g

  // The rows of X are staged K-chunk by K-chunk ((voxel tile, chunk) pairs in sequence).
  // The global loads of pair i + 1 are issued BEFORE the MFMAs of pair i and parked in
  // registers (px / psc / psh), so that between the two barriers of a pair only the LDS
  // writes remain: with one workgroup per CU nothing else would hide the HBM latency of
  // the staging (it used to cost as much as the MFMAs of a chunk).
  // Row u of a wave's share: FUSED -- a run of consecutive rows (rbeg + u), gathered
  // through src and z-scored at parking time with the (cell, voxel) scale / shift, which
  // are loaded per row (clamped addresses, no branches); plain -- rows 4 ks0 + 4 u + wave.
  struct Pos {
    int64_t vt;
    int ks0;
  };
  auto advance = [&](Pos q) {
    q.ks0 += A.ks;
    if (q.ks0 >= A.nk) {
      q.ks0 = 0;
      ++q.vt;
    }
    return q;
  };
  constexpr int PF = gram_pf(MC, FUSED);
  double px[PF], psc[FUSED ? PF : 1], psh[FUSED ? PF : 1];
  bool pvin = true;
  // FUSED: a row's two offsets come from the item's row table through the SCALAR cache (one
  // s_load_dwordx4 per row: the table is addressed as constant memory, the row index is wave-uniform),
  // and the three loads of a row share one per-lane voxel offset.  (Round 1 kept the source-row and
  // cell tables in vector registers and picked entries with v_readlane: ninety-odd instructions per
  // row with the 64-bit address arithmetic -- 4 700 per voxel tile and wave, more issue time than the
  // tile's MFMAs take.)
  typedef long long i64x4 __attribute__((ext_vector_type(4)));
  const int ntr = 4 * A.nk + GRAM_PF;
  const __attribute__((address_space(4))) i64x4 *rowtab =
      (const __attribute__((address_space(4))) i64x4 *)(uintptr_t)(A.rowtab + (FUSED ? (int64_t)item0 * ntr * 4 : 0));
  auto fetch = [&](Pos q) {
    const int ks1 = min(A.nk, q.ks0 + A.ks);
    const int64_t v = q.vt * TV + lane;
    const bool vin = v < A.p && q.vt < t_hi;
    const int64_t vc = min(v, A.p - 1);
    if (FUSED) {
      const int rpw = (4 * (ks1 - q.ks0) + WAVES - 1) / WAVES;
      const int rbeg = 4 * q.ks0 + wave * rpw;
      // rows past the wave's share or past n are fetched from a clamped row and never parked; voxels
      // past p (or a prefetch past the chunk) from a clamped voxel and parked as zeros (pvin)
      const uint32_t vo = (uint32_t)vc;                  // (p < 2^29: scalar base + 32-bit lane offset)
      // table entries in batches ahead of the loads that need them (one at a time, each scalar
      // load's round trip stood in front of its row's three vector loads)
      constexpr int TB = 5;
#pragma unroll
      for (int u0 = 0; u0 < PF; u0 += TB) {
        i64x4 t[TB];
#pragma unroll
        for (int u = 0; u < TB; ++u)
          if (u0 + u < PF) t[u] = rowtab[rbeg + u0 + u];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < TB; ++u)
          if (u0 + u < PF) {
            px[u0 + u] = ((const double *)(uintptr_t)t[u].x)[vo];
            psc[u0 + u] = ((const double *)(uintptr_t)t[u].y)[vo];
            psh[u0 + u] = ((const double *)(uintptr_t)t[u].z)[vo];
          }
        __builtin_amdgcn_sched_barrier(0);
      }
      pvin = vin;
    } else {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int row = 4 * q.ks0 + 4 * u + wave;
        const bool ok = row < 4 * ks1 && row < A.n;
        const double x = Xi[(int64_t)(ok ? row : 0) * A.ldx + vc];
        px[u] = (ok && vin) ? x : 0.0;
      }
    }
  };
  auto park = [&](Pos q) {
    const int ks1 = min(A.nk, q.ks0 + A.ks);
    if (FUSED) {
      const int rpw = (4 * (ks1 - q.ks0) + WAVES - 1) / WAVES;
      const int rbeg = 4 * q.ks0 + wave * rpw;
      const int rend = min(4 * ks1, rbeg + rpw);
      // the row's LDS slot: rows alternate between two lane offsets (xs_index swaps the 16-voxel
      // blocks of odd rows), so two base pointers chosen once by the first row's parity and a
      // compile-time row offset address all of them
      const int r0 = rbeg - 4 * q.ks0;
      double *even = Xs + xs_index(r0, lane), *odd = Xs + xs_index(r0 + 1, lane) - TV;
      const int nlive = min(rend, A.n) - rbeg;             // rows beyond are padding: zeros
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const double z = fma(px[u], psc[FUSED ? u : 0], psh[FUSED ? u : 0]);
        if (rbeg + u < rend) ((u & 1) ? odd : even)[u * TV] = (pvin && u < nlive) ? z : 0.0;
      }
    } else {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int row = 4 * q.ks0 + 4 * u + wave;
        if (row < 4 * ks1) Xs[xs_index(row - 4 * q.ks0, lane)] = px[u];
      }
    }
  };

  f64x4 D[B][MC];
  Pos cur{t_lo, 0};
  if (t_lo < t_hi) fetch(cur);
  while (cur.vt < t_hi) {
    const int ks0 = cur.ks0;
    const int ks1 = min(A.nk, ks0 + A.ks);
    if (ks0 == 0) {
#pragma unroll
      for (int b = 0; b < B; ++b)
#pragma unroll
        for (int mc = 0; mc < MC; ++mc) D[b][mc] = (f64x4){0.0, 0.0, 0.0, 0.0};
    }
    __syncthreads();                                    // previous X chunk fully consumed (and ops written)
    park(cur);
    __syncthreads();
    const Pos nxt = advance(cur);
    fetch(nxt);                                         // in flight during the MFMAs below
    {
      // software pipeline: the operands of k-step s + 1 are read from LDS before the
      // MFMAs of step s are issued (at one wave per SIMD nothing else hides the LDS
      // latency; left to itself the compiler reads and waits inside every step).  Two
      // register sets in turn, so that no copies sit between the loads and the MFMAs;
      // the last step of a chunk prefetches itself again (the next chunk is not staged yet).
      // (SPARSE: the group tests are wave-uniform bit tests of two scalar masks; a skipped group's
      // operand registers keep stale values that its skipped MFMAs never read)
      auto ld = [&](int sidx, double &a, double (&bv)[B][MC]) {
        a = Xs[(size_t)(sidx - ks0) * 4 * TV + xo];
        const bool lo_on = !SPARSE || ((act_lo >> sidx) & 1), hi_on = !SPARSE || ((act_hi >> sidx) & 1);
#pragma unroll
        for (int b = 0; b < B; ++b) {
          if (lo_on) {
#pragma unroll
            for (int mc = 0; mc < SL; ++mc) bv[b][mc] = ops[((size_t)(b * MC + mc) * A.nk + sidx) * 64 + lane];
          }
#pragma unroll
          for (int mc = SL; mc < SH; ++mc) bv[b][mc] = ops[((size_t)(b * MC + mc) * A.nk + sidx) * 64 + lane];
          if (hi_on) {
#pragma unroll
            for (int mc = SH; mc < MC; ++mc) bv[b][mc] = ops[((size_t)(b * MC + mc) * A.nk + sidx) * 64 + lane];
          }
        }
      };
      auto mm = [&](int sidx, double a, double (&bv)[B][MC]) {
        const bool lo_on = !SPARSE || ((act_lo >> sidx) & 1), hi_on = !SPARSE || ((act_hi >> sidx) & 1);
#pragma unroll
        for (int b = 0; b < B; ++b) {
          if (lo_on) {
#pragma unroll
            for (int mc = 0; mc < SL; ++mc) D[b][mc] = mfma_f64(a, bv[b][mc], D[b][mc]);
          }
#pragma unroll
          for (int mc = SL; mc < SH; ++mc) D[b][mc] = mfma_f64(a, bv[b][mc], D[b][mc]);
          if (hi_on) {
#pragma unroll
            for (int mc = SH; mc < MC; ++mc) D[b][mc] = mfma_f64(a, bv[b][mc], D[b][mc]);
          }
        }
      };
      double a0, a1, b0[B][MC], b1[B][MC];
      ld(ks0, a0, b0);
      int s = ks0;
      for (; s + 2 <= ks1; s += 2) {
        ld(s + 1, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        mm(s, a0, b0);
        const int s2 = min(s + 2, ks1 - 1);
        ld(s2, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        mm(s + 1, a1, b1);
      }
      if (s < ks1) mm(s, a0, b0);
    }
    if (ks1 == A.nk) {
#pragma unroll
      for (int b = 0; b < B; ++b) {
        int idx = 0;
#pragma unroll
        for (int m1 = 0; m1 < MC; ++m1)
#pragma unroll
          for (int m2 = m1; m2 < MC; ++m2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) G[b][idx] = mfma_f64(D[b][m1][r], D[b][m2][r], G[b][idx]);
            ++idx;
          }
      }
    }
    cur = nxt;
  }

  // ---- sum the four waves' Gram tiles through LDS and write the slab ----
  __syncthreads();
  double *red = smem;                                   // [B][NG][4][64], waves add in turn
  for (int w = 0; w < WAVES; ++w) {
    if (wave == w) {
#pragma unroll
      for (int b = 0; b < B; ++b)
#pragma unroll
        for (int i = 0; i < NG; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            double *dst = red + (((size_t)b * NG + i) * 4 + r) * 64 + lane;
            *dst = (w == 0 ? 0.0 : *dst) + G[b][i][r];
          }
    }
    __syncthreads();
  }
  for (int e = tid; e < B * MM * MM; e += 256) {
    const int b = e / (MM * MM);
    const int c1 = (e / MM) % MM;
    const int c2 = e % MM;
    const int item = item0 + b;
    if (item >= A.items) continue;
    // upper-triangle tile (m1 <= m2) holds G[c1][c2] for c1 in tile m1, c2 in tile m2
    int lo = c1, hi = c2;
    if ((c1 >> 4) > (c2 >> 4)) { lo = c2; hi = c1; }
    const int m1 = lo >> 4, m2 = hi >> 4;
    const int idx = m1 * MC - m1 * (m1 - 1) / 2 + (m2 - m1);
    const int rr = lo & 15, cc = hi & 15;               // row = (lane>>4) + 4*reg, col = lane&15
    const int src_lane = ((rr & 3) << 4) | cc;
    const int reg = rr >> 2;
    A.G_part[(((int64_t)chunk * A.items + item) * MM + c1) * MM + c2] =
        red[(((size_t)b * NG + idx) * 4 + reg) * 64 + src_lane];
  }
}

inline size_t gram_lds_bytes(int nk, int mc, int b, int ks) {
  size_t a = ((size_t)b * mc * nk * 64 + (size_t)ks * 4 * TV) * sizeof(double);
  size_t r = (size_t)b * (mc * (mc + 1) / 2) * 4 * 64 * sizeof(double);
  return a > r ? a : r;
}

// rows layout: tile t = item*MC + mc, lane (m = lane&15, kk = lane>>4) holds
// rows[item][16*mc + m][4*s + kk]
__global__ __launch_bounds__(256) void ops_rows_kernel(const double *rows, double *frag, int items,
                                                       int m, int n, int nk, int MC) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)items * MC * nk * 64;
  if (e >= total) return;
  const int lane = (int)(e & 63);
  const int s = (int)((e >> 6) % nk);
  const int64_t t = (e >> 6) / nk;
  const int item = (int)(t / MC);
  const int j = (int)(t % MC) * 16 + (lane & 15);
  const int i = 4 * s + (lane >> 4);
  frag[e] = (j < m && i < n) ? rows[((int64_t)item * m + j) * n + i] : 0.0;
}

// ---------------------------------------------------------------------------
// batched symmetric eigen-decomposition: one wavefront per matrix, cyclic
// two-sided Jacobi with a round-robin (tournament) ordering in LDS.
// Eigenvalues come back in descending order with matching eigenvector columns.
// ---------------------------------------------------------------------------
constexpr int EIG_MAX = 64;

// `init` (may be null): per item a k x k matrix V0 whose columns the rotations are applied to
// instead of the identity, so the result is V0 * J -- the accumulated basis of a refinement
// pass.  `relative`: the Gram was formed from rows that are already nearly orthogonal with
// graded norms (a refinement pass: G = D A D, A close to I, entry errors ~ eps d_i d_j), where
// two-sided Jacobi is RELATIVELY accurate (Demmel & Veselic) as long as no rotation is skipped
// for being small in absolute terms -- the noise clause below is then switched off.
__host__ __device__ inline int eigh_pitch(int k) { return (k | 1) + ((k & 1) ? 2 : 0); }      // odd, > k
inline size_t eigh_lds_bytes(int k) { return (size_t)2 * k * eigh_pitch(k) * sizeof(double); }

__global__ __launch_bounds__(64) void eigh_kernel(const double *Gsrc, int64_t item_stride, int ld,
                                                  int off, int k, int count, double *evals,
                                                  double *evecs, int max_sweeps, const double *init,
                                                  int relative) {
  // A and V in LDS, k rows of an odd pitch each -- sized by k at launch (eigh_lds_bytes): with the pitch of the largest
  // matrix (64 + 1: 66 KB per workgroup) a CU held two of these one-wave workgroups, and the kernel is bound by the
  // latency of its LDS round trips (at k = 38: 24 KB, six per CU)
  extern __shared__ double eig_sm[];
  __shared__ double cs[EIG_MAX], sn[EIG_MAX];
  __shared__ int pp[EIG_MAX], qq[EIG_MAX];
  const int item = blockIdx.x;
  if (item >= count) return;
  const int lane = threadIdx.x;
  const int LD = eigh_pitch(k);
  double *As = eig_sm, *Vs = eig_sm + k * LD;
  const double *G = Gsrc + (int64_t)item * item_stride + (int64_t)off * ld + off;
  const int kk = (k + 1) & ~1;            // players in the tournament (one dummy if k is odd)
  for (int e = lane; e < k * k; e += 64) {
    const int r = e / k, c = e % k;
    // symmetrise: the Gram is symmetric up to rounding of two summation orders
    As[r * LD + c] = 0.5 * (G[(int64_t)r * ld + c] + G[(int64_t)c * ld + r]);
    Vs[r * LD + c] = init ? init[(int64_t)item * k * k + e] : (r == c ? 1.0 : 0.0);
  }
  __syncthreads();
  for (int sweep = 0; sweep < max_sweeps; ++sweep) {
    // convergence: every off-diagonal entry is either small relative to its two
    // diagonal entries (1e-15 sqrt(a_pp a_qq)) or below the rounding noise of a
    // Gram with largest entry dmax (4 eps dmax).  The second clause stops the
    // sweeps on the null space of a rank-deficient Gram, whose entries are noise
    // of absolute size ~eps*dmax whatever their tiny diagonal says.
    double dmax = 0.0;
    for (int r = lane; r < k; r += 64) dmax = fmax(dmax, fabs(As[r * LD + r]));
    for (int sft = 32; sft > 0; sft >>= 1) dmax = fmax(dmax, __shfl_xor(dmax, sft));
    const double noise = relative ? 0.0 : 8.9e-16 * dmax;
    double worst = 0.0;
    for (int e = lane; e < k * k; e += 64) {
      const int r = e / k, c = e % k;
      if (r < c) {
        const double o = fabs(As[r * LD + c]);
        if (o > noise) {
          const double d = sqrt(fabs(As[r * LD + r] * As[c * LD + c]));
          worst = fmax(worst, d > 0.0 ? o / d : 1.0);
        }
      }
    }
    for (int sft = 32; sft > 0; sft >>= 1) worst = fmax(worst, __shfl_xor(worst, sft));
    if (worst < 1e-15) break;
    for (int step = 0; step < kk - 1; ++step) {
      // round-robin pairing: player 0 fixed, the others rotate
      if (lane < kk / 2) {
        int a = lane == 0 ? 0 : 1 + (lane - 1 + step) % (kk - 1);
        int b = 1 + (kk - 2 - lane + step) % (kk - 1);
        int p = min(a, b), q = max(a, b);
        double c = 1.0, s = 0.0;
        if (q < k) {
          const double apq = As[p * LD + q];
          const double app = As[p * LD + p], aqq = As[q * LD + q];
          if (fabs(apq) > 0.25 * noise && fabs(apq) >= 1e-17 * sqrt(fabs(app * aqq))) {
            const double tau = (aqq - app) / (2.0 * apq);
            const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
            c = 1.0 / sqrt(1.0 + t * t);
            s = t * c;
          }
        } else {
          q = p;                       // dummy opponent: identity
        }
        pp[lane] = p;
        qq[lane] = q;
        cs[lane] = c;
        sn[lane] = s;
      }
      __syncthreads();
      // columns: A <- A J, V <- V J   (lane = row)
      if (lane < k) {
        for (int i = 0; i < kk / 2; ++i) {
          const int p = pp[i], q = qq[i];
          if (p == q) continue;
          const double c = cs[i], s = sn[i];
          const double ap = As[lane * LD + p], aq = As[lane * LD + q];
          As[lane * LD + p] = c * ap - s * aq;
          As[lane * LD + q] = s * ap + c * aq;
          const double vp = Vs[lane * LD + p], vq = Vs[lane * LD + q];
          Vs[lane * LD + p] = c * vp - s * vq;
          Vs[lane * LD + q] = s * vp + c * vq;
        }
      }
      __syncthreads();
      // rows: A <- J^T A   (lane = column)
      if (lane < k) {
        for (int i = 0; i < kk / 2; ++i) {
          const int p = pp[i], q = qq[i];
          if (p == q) continue;
          const double c = cs[i], s = sn[i];
          const double ap = As[p * LD + lane], aq = As[q * LD + lane];
          As[p * LD + lane] = c * ap - s * aq;
          As[q * LD + lane] = s * ap + c * aq;
        }
      }
      __syncthreads();
    }
  }
  // sort descending (rank by counting) and write out
  double lam = lane < k ? As[lane * LD + lane] : 0.0;
  int rank = 0;
  if (lane < k) {
    for (int j = 0; j < k; ++j) {
      const double lj = As[j * LD + j];
      rank += (lj > lam) || (lj == lam && j < lane);
    }
    evals[(int64_t)item * k + rank] = lam;
    for (int r = 0; r < k; ++r) evecs[((int64_t)item * k + r) * k + rank] = Vs[r * LD + lane];
  }
}

// rows_out[item][off + j][:] = sum_i U[item][i][j] * rows_in[item][off + i][:]   (j < k): the
// operator rows of a decomposition's block expressed in its eigenvector basis -- the next
// refinement pass forms the Gram of THESE rows, and after the last pass they are the operator
// of the back-projection VS = rows_out @ X.  Rows outside the block are copied.
__global__ void rotate_rows_kernel(const double *U, const double *rows_in, double *rows_out, int m, int n,
                                   int off, int k) {
  const int item = blockIdx.y;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= m * n) return;
  const int r = e / n, c = e % n;
  const double *in = rows_in + (int64_t)item * m * n;
  double acc;
  if (r < off || r >= off + k) {
    acc = in[e];
  } else {
    const double *u = U + (int64_t)item * k * k;
    const int j = r - off;
    acc = 0.0;
    for (int i = 0; i < k; ++i) acc = fma(u[i * k + j], in[(int64_t)(off + i) * n + c], acc);
  }
  rows_out[(int64_t)item * m * n + e] = acc;
}

// Last step of the thin SVD on the device: singular values from the final pass's eigenvalues,
// deflation of null latent variables, and the two operators of the back-projection:
//   rows_out[i]     = live_i ? cur[i]         : 0      ->  (V s)^T = rows_out[:k]  @ X
//   rows_out[k + i] = live_i ? cur[i] / s_i   : 0      ->   V^T    = rows_out[k:]  @ X
// with cur = U^T rows (rotate_rows_kernel).  live_i: s_i > max(abs_tol, rel_tol * s_0).
__global__ void svd_finish_kernel(const double *lam, const double *cur, int k, int n, double abs_tol,
                                  double rel_tol, double *s_out, double *rows_out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= k * n) return;
  const int i = e / n;
  const double smax = sqrt(fmax(lam[0], 0.0));
  const double si = sqrt(fmax(lam[i], 0.0));
  const bool live = si > fmax(abs_tol, rel_tol * smax);
  const double x = cur[e];
  rows_out[e] = live ? x : 0.0;
  rows_out[(int64_t)k * n + e] = live ? x / si : 0.0;
  if (e % n == 0) s_out[i] = live ? si : 0.0;
}

}  // namespace plsr
