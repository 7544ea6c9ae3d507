// K1: batched operator projection for permutation / bootstrap resamples.
//
// One workgroup (4 waves, one per SIMD) owns a tile of PLSR_VOXEL_TILE = 64
// voxels.  X[:, tile] is staged once in LDS and every resample of the batch is
// streamed past it:
//
//     VS[c, v] = sum_i op_c[i] * X[i, v]        c = (resample, latent variable)
//
// as v_mfma_f64_16x16x4_f64 with M = 16 batch columns, N = 16 voxels, K = 4
// rows of X.  Accumulator layout (cdna guide section 3, f64 map):
//     D[row = (lane>>4) + 4*reg][col = lane & 15]
// so a lane's four registers are the four resamples of one quad (same latent
// variable) at one voxel: the sum over resamples for the bootstrap moments is
// register-local.  The column norms (s_hat^2) and the k2 x k product with the
// cell means of X (Tdistrib) contract over voxels, which sit on lanes; the
// accumulator tile is therefore bounced once through a per-wave LDS patch and
// re-read as an MFMA A operand (A[m = lane & 15][k = lane >> 4]).
//
// Reference arithmetic replaced: bootstrap_permutation.py:404-405 (perm),
// :617-634 and :695 (boot).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#ifndef PLSR_ABLATE
#define PLSR_ABLATE 0   // developer-only timing ablations (wrong results when non-zero)
#endif

namespace plsr {

typedef double f64x4 __attribute__((ext_vector_type(4)));

constexpr int TV = 64;         // voxels per workgroup
constexpr int NT = TV / 16;    // MFMA N-tiles per workgroup
constexpr int DT_LD = 18;      // padded row of the per-wave 16 x 16 transpose patch
constexpr int XM_LD = 72;      // padded row of the cell-mean tile
constexpr int WAVES = 4;
constexpr int MAX_PERIOD = 6;
constexpr int REG_NK_MIN = 4, REG_NK_MAX = 16;   // k-steps for which the X fragments of a wave fit its registers

struct ProjectArgs {
  const double *X;      // [n][ldx]
  int64_t ldx;
  int64_t p;
  int32_t n, nk;        // nk = ceil(n/4)
  const double *frag;   // [ntiles][nk][64]
  int32_t ntiles, k, kp, R, nquads;
  // outputs
  double *norm_part;    // [nvt][ntiles*16]
  double *T_part;       // [nvt][ntiles*16][k2]   (boot, k2 > 0)
  const double *ref;    // [p][k] or null
  const double *Xm;     // [k2][ldxm] or null
  int64_t ldxm;
  int32_t k2;
  double *S1, *S2;      // [column split][p][k] partial moment sums (overwritten)
  double *vs_dump;      // [R][p][k] or null
  // register-resident bootstrap kernel (LV-major layout)
  int32_t tpl, msplit;  // tiles per latent variable; splits of a latent variable's tiles (grid.y = k * msplit)
  const double *opsum;  // [4 nk][k] sum over the batch of the operators (S1 by linearity)
  double *sink;         // [64] scratch target of the lanes that have nothing to store (see K1r)
  // register-resident kernels, XCD-aware dispatch order (reg_tile_and_run): runs per voxel tile, voxel tiles
  int32_t xcd_runs;
  int64_t nvt;
};

// K1r / K1br: workgroup -> (voxel tile, run).  With A.xcd_runs set the grid is one-dimensional and
// workgroup ids go round the eight XCDs, so id % 8 picks the XCD: a voxel tile's runs are then neighbours
// on ONE XCD (run index fastest) and its X tile is fetched from HBM once, not once per run (at config 2 the
// re-reads were 0.57 GB of K1br's 0.67 GB of fetches, 0.6 of K1r's 0.7).  Returns false for the padding.
__device__ __forceinline__ bool reg_tile_and_run(const ProjectArgs &A, int64_t &vt, int &run, int &nrun) {
  if (A.xcd_runs > 0) {
    const int64_t wg = blockIdx.x, in_xcd = wg >> 3;
    nrun = A.xcd_runs;
    run = (int)(in_xcd % nrun);
    vt = (in_xcd / nrun) * 8 + (wg & 7);
    return vt < A.nvt;
  }
  vt = blockIdx.x;
  run = blockIdx.y;
  nrun = gridDim.y;
  return true;
}

__device__ __forceinline__ f64x4 mfma_f64(double a, double b, f64x4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// LDS image of the X tile: row i (0 .. 4*nk), 64 voxels; the four 16-voxel
// blocks of odd rows are stored pairwise swapped so that the two rows one
// ds_read_b64 half-wave touches fall in different halves of the 64 banks.
__device__ __forceinline__ int xs_index(int row, int v) {
  return row * TV + ((((v >> 4) ^ (row & 1)) << 4) | (v & 15));
}

// MODE 0: permutation (norms only); 1: bootstrap (moments, norms, T);
// 2: bootstrap that also materialises VS (tests, debug dict, observed blocks)
// NHT: halves (groups of four cells) of the second matrix, compile-time for the
// hot bootstrap instance (a run-time count in the unrolled MFMA loop costs ~15 %);
// -1 = take it from the arguments (dump mode), 0 = no second matrix.
// Largest workgroup an instance may be launched with (lds_fed_waves respects it).  The bootstrap
// instances keep 2 * PERIOD * NT moment accumulators in registers: with PERIOD >= 4 (and in dump
// mode) they need more than the 168 VGPRs a twelve-wave workgroup leaves a wave, and used to
// spill 32-250 bytes per lane; eight waves (256 VGPRs) hold them.
template <int PERIOD, int MODE>
constexpr int project_max_threads() {
  return MODE == 0 ? 1024 : ((MODE == 1 && PERIOD <= 3) ? 768 : 512);
}
inline int project_max_waves(int period, int mode) {
  return mode == 0 ? 16 : ((mode == 1 && period <= 3) ? 12 : 8);
}

template <int PERIOD, int MODE, int NHT>
__global__ __launch_bounds__((project_max_threads<PERIOD, MODE>()), (MODE == 1 && PERIOD <= 3) ? 3 : 2) void project_kernel(ProjectArgs A) {
  constexpr bool BOOT = MODE != 0;
  constexpr bool DUMP = MODE == 2;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // four waves per workgroup, or eight and more when the X tile leaves room for one or two
  // workgroups per CU only (lds_fed_waves)
  const int NW = blockDim.x >> 6;
  const int col = lane & 15;
  const int g = lane >> 4;
  const int64_t vt = blockIdx.x;
  const int64_t v0 = vt * TV;
  const int nrows = A.nk * 4;

  double *Xs = smem;                       // nrows * 64
  double *Dt = smem + (size_t)nrows * TV + (size_t)wave * 16 * DT_LD;

  // ---- stage X[:, v0 : v0+64] (zero padded in both directions) ----
  for (int r0 = 0; r0 < nrows; r0 += 4 * NW) {
    double tmp[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int row = r0 + u * NW + wave;
      const int64_t v = v0 + lane;
      tmp[u] = (row < A.n && v < A.p) ? A.X[(int64_t)row * A.ldx + v] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int row = r0 + u * NW + wave;
      if (row < nrows) Xs[xs_index(row, lane)] = tmp[u];
    }
  }

  // per-lane B-operand offsets into Xs for k-step 0: row = g, voxel = 16*nt + col
  int xo[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) xo[nt] = xs_index(g, nt * 16 + col);

  // second matrix (cell means of X) as B operand of the voxel contraction:
  // B[k = v = 4*s + g][n = c' = col]
  // for v_mfma_f64_4x4x4_4b_f64 (four independent 4x4x4 blocks per instruction;
  // lane = 16 k + 4 b + x: A_b[i = x][k], B_b[k][j = x], D_b[i][j] at lane 16 i + 4 b + j;
  // probed in microbench/mfma_f64_4x4.hip).  Block b = four batch columns, j = four
  // cells of "half" h: with k2 <= 12 cells the 16x16x4 shape would waste most of
  // its N dimension, the 4x4 shape costs 17 instead of 64 cycles per half.
  // B_b[k][j] = Xm[4 h + j][v0 + 4 step + k] for every b: the four b-lanes read the
  // same LDS word (broadcast) and the padded row (XM_LD = 72) puts the four cells of a
  // half 16 banks apart, so the read is conflict free.  Shared by the four waves.
  const int nh = NHT >= 0 ? NHT : (A.k2 + 3) / 4;                         // halves of 4 cells
  double *XmS = smem + (size_t)nrows * TV + (size_t)NW * 16 * DT_LD;   // [4 nh cells][XM_LD]
  // shift of the streaming moments (observed V*s), [kp][64 voxels], zero for padding
  double *RfS = XmS + (size_t)nh * 4 * XM_LD;
  double s1[PERIOD][NT], s2[PERIOD][NT];
  if (BOOT) {
    for (int e = tid; e < A.kp * TV; e += blockDim.x) {
      const int j = e >> 6;
      const int64_t v = v0 + (e & 63);
      RfS[e] = (A.ref != nullptr && j < A.k && v < A.p) ? A.ref[v * A.k + j] : 0.0;
    }
    for (int cell = wave; cell < nh * 4; cell += NW) {
      const int64_t v = v0 + lane;
      XmS[cell * XM_LD + lane] =
          (A.Xm != nullptr && cell < A.k2 && v < A.p) ? A.Xm[(int64_t)cell * A.ldxm + v] : 0.0;
    }
#pragma unroll
    for (int sl = 0; sl < PERIOD; ++sl) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        s1[sl][nt] = 0.0;
        s2[sl][nt] = 0.0;
      }
    }
  }
  __syncthreads();

  const int64_t C = (int64_t)A.ntiles * 16;
  const int nk = A.nk;
  // blockIdx.y splits the batch tiles (in whole wave x period groups) so that the
  // grid quantises well on 256 CUs; each split owns its own moment partials.
  // Inside a split every wave takes a CONTIGUOUS run of tiles (a multiple of
  // PERIOD, so the slot -> latent-variable map holds): its operator fragments
  // are then one linear stream in memory.
  const int cs = blockIdx.y;
  const int group = NW * PERIOD;
  const int gps = ((A.ntiles + group - 1) / group + gridDim.y - 1) / gridDim.y;
  const int t_begin = cs * gps * group;
  const int t_end = min(A.ntiles, t_begin + gps * group);
  const int run = ((max(t_end - t_begin, 0) + group - 1) / group) * PERIOD;   // tiles per wave
  const int w_lo = min(t_end, t_begin + wave * run);
  const int w_hi = min(t_end, w_lo + run);
  const int rem = nk & 3;

  // Software pipeline of the main contraction.  `ring` holds this wave's
  // operator fragments (MFMA A operand) four k-steps ahead of use.  Because the
  // wave's tiles are contiguous the prefetch simply runs on into the next tile
  // (the buffer carries four k-steps of padding at its end), so the L2 latency of
  // a fragment load is always covered by 16 MFMAs.  `bn` holds the X fragments
  // (B operand, from LDS) one k-step ahead.  The loop body has no branches.
  double ring[4];
  int phase = 0;              // ring slot holding the current tile's first k-step
  double bn[NT];
  {
    const double *ap0 = A.frag + ((size_t)w_lo * nk) * 64 + lane;
#pragma unroll
    for (int u = 0; u < 4; ++u) ring[u] = ap0[(size_t)u * 64];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bn[nt] = Xs[xo[nt]];
  }

  for (int base = w_lo; base < w_hi; base += PERIOD) {
#pragma unroll
    for (int sl = 0; sl < PERIOD; ++sl) {
      const int t = base + sl;
      if (t >= w_hi) break;

      // ---------------- main contraction ----------------
      f64x4 acc[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = (f64x4){0.0, 0.0, 0.0, 0.0};

      const double *ap = A.frag + ((size_t)t * nk) * 64 + lane;

      auto step = [&](int s, double a) {
        double b[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[nt] = bn[nt];
        const int s1 = (s + 1 == nk) ? 0 : s + 1;            // same X tile for every batch tile
        const double *xr = Xs + (size_t)s1 * 4 * TV;
#if !(PLSR_ABLATE & 4)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bn[nt] = xr[xo[nt]];
#endif
        // keep the next step's LDS reads (and the fragment prefetch issued by the
        // caller) ahead of this step's MFMAs; the scheduler otherwise sinks them
        // below and exposes the LDS latency at every k-step
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma_f64(a, b[nt], acc[nt]);
      };
      auto fetch = [&](int s) -> double {      // fragment of k-step s counted from this tile (runs on linearly)
#if PLSR_ABLATE & 2
        return 1.0 + s;
#else
        return ap[(size_t)s * 64];
#endif
      };

      // nk is rarely a multiple of four, so the ring slot of a tile's first k-step
      // ("phase") moves from tile to tile.  Rotating the ring registers back would
      // need every outstanding fragment load returned (s_waitcnt vmcnt(0) once per
      // tile); instead the k-loop exists in four variants with static slot numbers.
      auto kloop = [&](auto ph) {
        constexpr int P = decltype(ph)::value;
        int s = 0;
        for (; s + 4 <= nk; s += 4) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            // refill the slot after the MFMAs that read it have been issued: the
            // load can then target the same register
            step(s + u, ring[(P + u) & 3]);
            ring[(P + u) & 3] = fetch(s + u + 4);
          }
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          if (u < rem) {
            step(s + u, ring[(P + u) & 3]);
            ring[(P + u) & 3] = fetch(s + u + 4);
          }
        }
      };
      switch (phase) {
        case 0: kloop(std::integral_constant<int, 0>{}); break;
        case 1: kloop(std::integral_constant<int, 1>{}); break;
        case 2: kloop(std::integral_constant<int, 2>{}); break;
        default: kloop(std::integral_constant<int, 3>{}); break;
      }
      phase = (phase + rem) & 3;

      // quad / resample bookkeeping of this lane group
      const int q = 4 * t + g;
      const int j = q % A.kp;
      const int bg = q / A.kp;
      const bool live = (q < A.nquads) && (j < A.k);

      if (BOOT) {
        // ---------------- streaming moments over resamples ----------------
        const int nvalid = live ? min(4, A.R - 4 * bg) : 0;   // resamples of this quad that exist
        double rf[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) rf[nt] = RfS[j * TV + nt * 16 + col];
#if PLSR_ABLATE & 8
        if (nvalid == 77) {
          s1[sl][0] += acc[0][0];
          s2[sl][0] += acc[1][0];
        } else if (nvalid == 78) {
#else
        if (__builtin_expect(__all(nvalid == 4), 1)) {
#endif
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const double d = acc[nt][r] - rf[nt];
              s1[sl][nt] += d;
              s2[sl][nt] = fma(d, d, s2[sl][nt]);
            }
          }
        } else {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const double d = (r < nvalid) ? acc[nt][r] - rf[nt] : 0.0;
              s1[sl][nt] += d;
              s2[sl][nt] = fma(d, d, s2[sl][nt]);
            }
          }
        }
        if (DUMP && A.vs_dump != nullptr && live) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const int64_t v = v0 + nt * 16 + col;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int b = 4 * bg + r;
              if (b < A.R && v < A.p) A.vs_dump[((int64_t)b * A.p + v) * A.k + j] = acc[nt][r];
            }
          }
        }
      }

#if PLSR_ABLATE & 1
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) asm volatile("" ::"v"(acc[nt]));
      continue;
#endif
      // ---------------- voxel contractions via the transpose patch ----------------
      // one 16-voxel block at a time through a 16 x 18 (padded) per-wave patch:
      // written in accumulator layout, read back as A[m = col][k = g]
      double nsq = 0.0;
      double accT[4] = {0.0, 0.0, 0.0, 0.0};                 // one 4x4x4_4b accumulator per half
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
#if PLSR_ABLATE & 64
        double av[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) av[r] = acc[nt][r];
#else
#pragma unroll
        for (int r = 0; r < 4; ++r) Dt[(g + 4 * r) * DT_LD + col] = acc[nt][r];
        __builtin_amdgcn_wave_barrier();
        double av[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) av[s] = Dt[col * DT_LD + 4 * s + g];
        __builtin_amdgcn_wave_barrier();
#endif
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          nsq = fma(av[s], av[s], nsq);
          if (BOOT && !(PLSR_ABLATE & 16)) {
#pragma unroll
            for (int h = 0; h < 4; ++h)
              if (h < nh)
                accT[h] = __builtin_amdgcn_mfma_f64_4x4x4f64(
                    av[s], XmS[(4 * h + (lane & 3)) * XM_LD + 16 * nt + 4 * s + g], accT[h], 0, 0, 0);
          }
        }
      }
      nsq += __shfl_xor(nsq, 16);
      nsq += __shfl_xor(nsq, 32);
#if !(PLSR_ABLATE & 32)
      {
        // unconditional store (idle lanes write a sink): a store under a branch makes
        // the compiler wait for vmcnt(0) where the fragment ring is used next
        double *dst = g == 0 ? A.norm_part + (vt * C + (int64_t)t * 16 + col) : A.sink + lane;
        *dst = nsq;
      }
      if (BOOT) {
        // D_b[i][j] sits at lane 16 i + 4 b + j: batch column c = 4 b + i, cell = 4 h + j
        const int c = 4 * ((lane & 15) >> 2) + (lane >> 4);
        double *tp = A.T_part + (vt * C + (int64_t)t * 16 + c) * A.k2;     // never dereferenced when k2 = 0
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          const int cell = 4 * h + (lane & 3);
          if (h < nh) {
            double *dst = cell < A.k2 ? tp + cell : A.sink + lane;
            *dst = accT[h];
          }
        }
      }
#endif
    }
  }

  if (BOOT && !(PLSR_ABLATE & 128)) {
    // ---- fold the per-wave, per-quad-slot moment registers into S1/S2 ----
    // scratch[wave][slot][g][v]   (aliases the X tile; all waves are done with it)
    double *red = smem;
    for (int pass = 0; pass < 2; ++pass) {
      __syncthreads();
#pragma unroll
      for (int sl = 0; sl < PERIOD; ++sl) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          red[((wave * PERIOD + sl) * 4 + g) * TV + nt * 16 + col] = pass == 0 ? s1[sl][nt] : s2[sl][nt];
      }
      __syncthreads();
      double *out = (pass == 0 ? A.S1 : A.S2) + (int64_t)cs * A.p * A.k;
      for (int e = tid; e < TV * A.k; e += blockDim.x) {
        const int vl = e / A.k;
        const int j = e % A.k;
        const int64_t v = v0 + vl;
        if (v >= A.p) continue;
        double sum = 0.0;
        for (int w = 0; w < NW; ++w) {
          for (int qq = j; qq < 4 * PERIOD; qq += A.kp)
            sum += red[((w * PERIOD + (qq >> 2)) * 4 + (qq & 3)) * TV + vl];
        }
        out[v * A.k + j] = sum;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// K1r: permutation projection with the X fragments resident in REGISTERS.
//
// One wave owns 64 voxels for a run of batch tiles.  The product is taken in the
// transposed orientation, D[voxel][batch column]: the X fragment is the MFMA A
// operand (A[m = voxel][k = row]; the same lane contents the LDS-fed kernel reads
// as its B operand) and the operator fragment the B operand (same buffer,
// unchanged).  With nk <= 16 the 4 x nk X fragments of the wave fit its register
// file (<= 128 VGPRs), so the k-loop reads nothing but the operator fragments,
// and those a whole tile ahead (fr[]): the MFMA pipe runs at the register-fed
// rate (microbench/mfma_f64_regB.hip: 74-75 TFLOP/s at 1-2 waves per SIMD,
// against 58-71 for one ds_read per MFMA).  In this orientation the column
// norms are sums over the accumulator's ROW index: 16 FMAs per lane and two
// cross-lane adds -- no LDS transpose.  No LDS, no barriers, workgroup = wave.
// (Bootstrap keeps the LDS-fed kernel: its moment sums need the other
// orientation, and its extra registers do not fit beside the X fragments.)
template <int NK>
__global__ __launch_bounds__(64, 2) void project_perm_reg_kernel(ProjectArgs A) {
  const int lane = threadIdx.x;
  const int col = lane & 15;
  const int g = lane >> 4;
  int64_t vt;
  int run, nrun;
  if (!reg_tile_and_run(A, vt, run, nrun)) return;
  const int64_t v0 = vt * TV;

  double xa[NK][NT];
#pragma unroll
  for (int s = 0; s < NK; ++s)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int row = 4 * s + g;
      const int64_t v = v0 + nt * 16 + col;
      xa[s][nt] = (row < A.n && v < A.p) ? A.X[(int64_t)row * A.ldx + v] : 0.0;
    }

  const int per = (A.ntiles + nrun - 1) / nrun;
  const int t_lo = run * per;
  const int t_hi = min(A.ntiles, t_lo + per);
  if (t_lo >= t_hi) return;
  const int64_t C = (int64_t)A.ntiles * 16;

  double fr[NK];
  {
    const double *ap = A.frag + ((size_t)t_lo * NK) * 64 + lane;
#pragma unroll
    for (int s = 0; s < NK; ++s) fr[s] = ap[(size_t)s * 64];
  }
  for (int t = t_lo; t < t_hi; ++t) {
    // next tile's fragments (the last tile of the buffer re-reads itself)
    const double *an = A.frag + ((size_t)min(t + 1, A.ntiles - 1) * NK) * 64 + lane;
    f64x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = (f64x4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < NK; ++s) {
      const double b = fr[s];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma_f64(xa[s][nt], b, acc[nt]);
      fr[s] = an[(size_t)s * 64];
      // keep the refill where it is written: left alone the scheduler sinks most of the
      // fifteen loads to the end of the tile, three k-steps before their first use
      __builtin_amdgcn_sched_barrier(0);
    }
    // acc[nt][r] = VS[batch column col][voxel 16 nt + g + 4 r]
    double q = 0.0;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) q = fma(acc[nt][r], acc[nt][r], q);
    q += __shfl_xor(q, 16);
    q += __shfl_xor(q, 32);
    // every lane stores (the idle ones into a sink): a store under a branch would
    // make the next tile's first MFMA wait for vmcnt(0), i.e. for this store's
    // acknowledgement, because the fragment loads share the counter
    double *dst = g == 0 ? A.norm_part + (vt * C + (int64_t)t * 16 + col) : A.sink + lane;
    *dst = q;
  }
}

// ---------------------------------------------------------------------------
// K1br: bootstrap projection with register-resident X fragments.
//
// Same orientation as K1r (D[voxel][batch column]); the batch is in LV-major
// order (tile = 16 resamples of one latent variable) and a wave's run of tiles
// stays inside one latent variable j (grid.y = k * msplit), so that
//   * sum_b VS^2 (second moment) accumulates per lane in 16 registers over the
//     whole run and is folded over the 16 column lanes once, at the end;
//   * sum_b VS (first moment) needs no accumulation at all: it is linear,
//     X^T (sum_b Op_b), formed from the wave's own X registers and the summed
//     operator (A.opsum) by the split that owns the variable's first tiles;
//   * the column norms are sums over the accumulator's row index (registers + two
//     cross-lane adds) and the product with the cell means (Tdistrib) takes the
//     accumulator registers directly as the B operand of v_mfma_f64_4x4x4_4b
//     (B_b[k][j]: k = voxel, j = batch column) -- no LDS transpose anywhere.
// Moments are plain sums (zero-padded columns contribute nothing); the shift by
// the observed VS is applied at the merge (moment_shift_merge_kernel).  Precision: expanding
// sum (x - ref)^2 = sum x^2 - 2 ref sum x + R ref^2 cancels (ref / sd)^2 -- the squared bootstrap
// ratio of the voxel -- of the 2^53: a voxel with |boot ratio| = 30 keeps 1e-13 relative accuracy in
// std_errs (the parity tolerance is 1e-9); the LDS-fed kernel and K4a / K4b's merge do the same or
// accumulate about ref directly.  A voxel whose bootstrap values are constant to 1e-8 of their size
// would lose everything -- so would the reference's np.std within a few digits.
// LDS (per wave = per workgroup): the 64 voxels of the cell means [4 nh][XM_LD] and the
// second-moment accumulators [16][64] (lane-private slots; in registers they
// pushed the nk >= 12 instances into scratch).
// NHT: number of four-cell groups of the second matrix at compile time (-1: from the
// arguments); with it the tile loop has no branch at all, which is what lets the
// compiler keep the fragment loads in flight across the loop's back-edge.
// (occupancy: two waves per SIMD up to NK = 15 -- 252 VGPRs at NK = 15; the sixteen-step instance
// (n = 61..64) and the dump instances from NK = 12 on do not fit 256 and spilled 12-164 bytes per
// lane: they take one wave per SIMD, where a register-fed fp64 MFMA loop still runs at 74 of
// the 77 TFLOP/s, microbench/mfma_f64_regB.hip)
template <int NK, bool DUMP, int NHT>
__global__ __launch_bounds__(64, (NK >= 16 || (DUMP && NK >= 12)) ? 1 : 2) void project_boot_reg_kernel(ProjectArgs A) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x;
  const int col = lane & 15;
  const int g = lane >> 4;
  int64_t vt;
  int run, nrun;
  if (!reg_tile_and_run(A, vt, run, nrun)) return;
  const int64_t v0 = vt * TV;
  const int j = run / A.msplit;                // latent variable of this run
  const int mi = run % A.msplit;
  const int nh = NHT >= 0 ? NHT : (A.k2 + 3) / 4;

  double xa[NK][NT];
#pragma unroll
  for (int s = 0; s < NK; ++s)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int row = 4 * s + g;
      const int64_t v = v0 + nt * 16 + col;
      xa[s][nt] = (row < A.n && v < A.p) ? A.X[(int64_t)row * A.ldx + v] : 0.0;
    }
  for (int cell = 0; cell < 4 * nh; ++cell) {
    const int64_t v = v0 + lane;
    smem[cell * XM_LD + lane] = (cell < A.k2 && v < A.p) ? A.Xm[(int64_t)cell * A.ldxm + v] : 0.0;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();

  const int per = (A.tpl + A.msplit - 1) / A.msplit;
  const int t_lo = j * A.tpl + min(A.tpl, mi * per);
  const int t_hi = j * A.tpl + min(A.tpl, mi * per + per);
  const int64_t C = (int64_t)A.ntiles * 16;

  double *s2 = smem + 4 * nh * XM_LD + lane;          // s2[(4 nt + r) * 64]
#pragma unroll
  for (int i = 0; i < 4 * NT; ++i) s2[i * 64] = 0.0;

  double fr[NK];
  {
    const double *ap = A.frag + ((size_t)min(t_lo, A.ntiles - 1) * NK) * 64 + lane;
#pragma unroll
    for (int s = 0; s < NK; ++s) fr[s] = ap[(size_t)s * 64];
  }
  for (int t = t_lo; t < t_hi; ++t) {
    const double *an = A.frag + ((size_t)min(t + 1, A.ntiles - 1) * NK) * 64 + lane;
    f64x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = (f64x4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < NK; ++s) {
      const double b = fr[s];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma_f64(xa[s][nt], b, acc[nt]);
      fr[s] = an[(size_t)s * 64];
    }
    // acc[nt][r] = VS_b[voxel 16 nt + g + 4 r][j],  b = 16 (t - j tpl) + col
    // (scheduling barriers keep the epilogue's LDS operands from being hoisted all
    // at once next to the resident X fragments: that spilled)
    // column norms first; the second-moment LDS atomics come LAST in the epilogue: LDS
    // operations of a wave complete in order, so every later s_waitcnt lgkmcnt -- the two
    // cross-lane adds below, the cell-mean reads of the T product -- would also wait for
    // sixteen ds_add_f64 issued in front of them
    // (the first cell-mean operands of the T product are read here, so that their LDS
    // latency passes under the norm computation)
    double xn[NT];
    {
      const double *xm0 = smem + (lane & 3) * XM_LD + g;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) xn[nt] = xm0[16 * nt];
    }
    double q = 0.0;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) q = fma(acc[nt][r], acc[nt][r], q);
    q += __shfl_xor(q, 16);
    q += __shfl_xor(q, 32);
    // stores are unconditional (idle lanes write a sink, see K1r) and fixed in number,
    // so the compiler can count them and the fragment loads apart (no vmcnt(0) per tile)
    {
      double *dst = g == 0 ? A.norm_part + (vt * C + (int64_t)t * 16 + col) : A.sink + lane;
      *dst = q;
    }
    double Tout[4] = {0.0, 0.0, 0.0, 0.0};
    // (straight-line code under uniform guards, not a loop: a child loop in the tile
    // loop makes the compiler wait for vmcnt(0) at the tile loop's header)
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      if (h >= nh) continue;
#if PLSR_ABLATE & 2048                   // dev: no T product (and none of its cell-mean reads)
      continue;
#endif
      // T[cell 4 h + i][column 4 b + jj] lands at lane 16 i + 4 b + jj = (g = i, col).
      // Four independent accumulation chains (one per voxel tile): a single chain of
      // sixteen dependent 4x4x4 MFMAs would expose the MFMA latency sixteen times.
      // The cell-mean operands of the next step (r + 1, or the first of group h + 1) are
      // read while this step multiplies.
      double aT[NT] = {0.0, 0.0, 0.0, 0.0};
      const double *xm = smem + (4 * h + (lane & 3)) * XM_LD + g;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double xc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) xc[nt] = xn[nt];
        if (r < 3) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) xn[nt] = xm[16 * nt + 4 * (r + 1)];
        } else if (h + 1 < nh) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) xn[nt] = xm[4 * XM_LD + 16 * nt];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          aT[nt] = __builtin_amdgcn_mfma_f64_4x4x4f64(xc[nt], acc[nt][r], aT[nt], 0, 0, 0);
      }
      Tout[h] = (aT[0] + aT[1]) + (aT[2] + aT[3]);
    }
    {
      double *tp = A.T_part + (vt * C + (int64_t)t * 16 + col) * A.k2;     // never dereferenced when k2 = 0
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        if (NHT >= 0 && h >= NHT) continue;      // (compile-time: the number of stores stays fixed)
        const int cell = 4 * h + g;
        double *dst = cell < A.k2 ? tp + cell : A.sink + lane;
        *dst = Tout[h];
      }
    }
    // second moment: ds_add_f64 without return on lane-private slots -- the order of
    // additions is program order (deterministic), and no register or wait is spent on it
    // (nothing reads LDS again before the next tile's epilogue)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#if PLSR_ABLATE & 512                    // dev: no second moment at all
        (void)s2;
#elif PLSR_ABLATE & 1024                 // dev: plain LDS stores instead of the atomics (wrong sums)
        s2[(4 * nt + r) * 64] = acc[nt][r] * acc[nt][r];
#else
        __builtin_amdgcn_ds_atomic_fadd_f64((__attribute__((address_space(3))) double *)(s2 + (4 * nt + r) * 64),
                                            acc[nt][r] * acc[nt][r]);
#endif
      }
    }
    if (DUMP) {
      const int64_t b = (int64_t)(t - j * A.tpl) * 16 + col;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t v = v0 + nt * 16 + g + 4 * r;
          if (b < A.R && v < A.p) A.vs_dump[(b * A.p + v) * A.k + j] = acc[nt][r];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- second moment: fold the 16 column lanes, one store per (voxel, j) ----
  double *o2 = A.S2 + (int64_t)mi * A.p * A.k;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double x = s2[(4 * nt + r) * 64];
      x += __shfl_xor(x, 1);
      x += __shfl_xor(x, 2);
      x += __shfl_xor(x, 4);
      x += __shfl_xor(x, 8);
      const int64_t v = v0 + nt * 16 + g + 4 * r;
      if (col == 0 && v < A.p) o2[v * A.k + j] = x;
    }
  // ---- first moment by linearity (the split that owns the variable's first tiles) ----
  if (mi == 0) {
    double a1[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) a1[nt] = 0.0;
#pragma unroll
    for (int s = 0; s < NK; ++s) {
      const double os = A.opsum[(int64_t)(4 * s + g) * A.k + j];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) a1[nt] = fma(xa[s][nt], os, a1[nt]);
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      double x = a1[nt];
      x += __shfl_xor(x, 16);
      x += __shfl_xor(x, 32);
      const int64_t v = v0 + nt * 16 + col;
      if (g == 0 && v < A.p) A.S1[v * A.k + j] = x;
    }
  }
}

// opsum[i][j] = sum over the batch of Op_b[i][j], from the LV-major fragments
// (frag[t][s][lane]: lane (g, m) holds Op of resample 16 (t % tpl) + m, variable t / tpl, row 4 s + g).
// One workgroup per (k-step s, variable j); its four waves split the variable's tiles.
__global__ __launch_bounds__(256) void opsum_kernel(const double *frag, double *opsum, int nk, int k, int tpl) {
  __shared__ double part[WAVES][4];
  const int s = blockIdx.x, j = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double a = 0.0;
  for (int tt = wave; tt < tpl; tt += WAVES) a += frag[((size_t)(j * tpl + tt) * nk + s) * 64 + lane];
  a += __shfl_xor(a, 1);
  a += __shfl_xor(a, 2);
  a += __shfl_xor(a, 4);
  a += __shfl_xor(a, 8);
  if ((lane & 15) == 0) part[wave][lane >> 4] = a;
  __syncthreads();
  if (threadIdx.x < 4) {
    const int g = threadIdx.x;
    opsum[(int64_t)(4 * s + g) * k + j] = (part[0][g] + part[1][g]) + (part[2][g] + part[3][g]);
  }
}

// S1 += P1 - cnt * ref,  S2 += sum_c P2[c] - 2 ref P1 + cnt ref^2   (plain sums -> shifted sums)
__global__ __launch_bounds__(256) void moment_shift_merge_kernel(double *S1, double *S2, const double *P1,
                                                                const double *P2, const double *ref,
                                                                int64_t count, int nsplit, double cnt) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= count) return;
  const double a = P1[e];
  double b = P2[e];
  for (int c = 1; c < nsplit; ++c) b += P2[(int64_t)c * count + e];
  const double r = ref != nullptr ? ref[e] : 0.0;
  S1[e] += a - cnt * r;
  S2[e] += fma(r, fma(cnt, r, -2.0 * a), b);
}

inline size_t project_lds_bytes(int nk, int period, bool boot, int nh, int kp, int nw = WAVES) {
  size_t a = ((size_t)nk * 4 * TV + (size_t)nw * 16 * DT_LD + (boot ? nh * 4 * XM_LD + kp * TV : 0)) *
             sizeof(double);
  size_t b = boot ? (size_t)nw * period * 4 * TV * sizeof(double) : 0;
  return a > b ? a : b;
}

// Waves per workgroup of the LDS-fed kernel: when the X tile alone (n x 64 doubles) takes
// more than half of the CU's 160 KB only one workgroup is resident, and four waves would
// leave each SIMD with a single wave (LDS-fed fp64 MFMA: 58 TFLOP/s at one wave per SIMD,
// 66 at two, microbench/mfma_f64_data) -- eight waves then, if the per-wave patches still fit.
inline int lds_fed_waves(int nk, int period, bool boot, int nh, int kp, int mode = -1) {
  if (mode < 0) mode = boot ? 1 : 0;
  const int cap = project_max_waves(period, mode);
  const size_t tile = (size_t)nk * 4 * TV * sizeof(double);
  // (the permutation instance needs 94 VGPRs only: from two workgroups per CU on -- X tile above
  // 36 KB, n > 72 -- eight waves each put four waves on a SIMD: -2.5 % at n = 120)
  if (tile <= (size_t)(boot ? 72 : 36) * 1024) return WAVES;
  if (tile > 72 * 1024) {
    // one workgroup per CU: as many waves as the registers (bootstrap 165 VGPRs: three per
    // SIMD; permutation: four) and the per-wave patches in LDS allow -- config 5 (n = 240):
    // ten / fourteen waves, another 2 % over eight
    for (int nw = cap; nw > 2 * WAVES; nw -= 2)
      if (project_lds_bytes(nk, period, boot, nh, kp, nw) <= 160 * 1024) return nw;
  }
  return project_lds_bytes(nk, period, boot, nh, kp, 2 * WAVES) <= 160 * 1024 ? 2 * WAVES : WAVES;
}

// ---------------------------------------------------------------------------
// operator fragments
// ---------------------------------------------------------------------------
struct OpsArgs {
  const int32_t *inds;   // [R][n]            (select mode)
  const double *M;       // [n][k]            (select mode)
  const double *cols;    // [R][k][n]         (dense mode)
  double *frag;          // [ntiles][nk][64]
  int32_t n, nk, k, kp, R, nquads, ntiles;
  int32_t tpl;           // > 0: LV-major layout, tiles per latent variable
};

template <bool DENSE>
__global__ __launch_bounds__(256) void ops_kernel(OpsArgs A) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)A.ntiles * A.nk * 64;
  if (e >= total) return;
  const int lane = (int)(e & 63);
  const int s = (int)((e >> 6) % A.nk);
  const int t = (int)((e >> 6) / A.nk);
  const int m = lane & 15;
  const int i = 4 * s + (lane >> 4);
  int q, j, b;
  if (A.tpl > 0) {               // LV-major: tile t = 16 resamples of latent variable t / tpl
    q = 0;
    j = t / A.tpl;
    b = (t % A.tpl) * 16 + m;
  } else {                       // quad layout
    q = 4 * t + (m & 3);
    j = q % A.kp;
    b = 4 * (q / A.kp) + (m >> 2);
  }
  double val = 0.0;
  if (q < A.nquads && j < A.k && b < A.R && i < A.n) {
    if (DENSE) {
      val = A.cols[((int64_t)b * A.k + j) * A.n + i];
    } else {
      const int32_t *ib = A.inds + (int64_t)b * A.n;
      for (int r = 0; r < A.n; ++r)
        if (ib[r] == i) val += A.M[(int64_t)r * A.k + j];
    }
  }
  A.frag[e] = val;
}

// Behaviour-PLS operators straight from the z-scored behaviour block
// (class_functions.py:240-242 folded with `@ U`, bootstrap_permutation.py:404):
//   Op_b[i][j] = sum_beh Yz[b][i][beh] * U[cell(i) * nb + beh][j]
// -- the R x k x n operator columns are never built on the host.
struct OpsBehArgs {
  const double *Yz;        // [R][n][nb] per-cell z-scored behaviour of every resample
  const double *U;         // [ncell * nb][k]
  const int32_t *rowcell;  // [n]
  int32_t nb;
  OpsArgs o;               // layout fields (inds / M / cols unused)
};

__global__ __launch_bounds__(256) void ops_behaviour_kernel(OpsBehArgs B) {
  const OpsArgs &A = B.o;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)A.ntiles * A.nk * 64;
  if (e >= total) return;
  const int lane = (int)(e & 63);
  const int s = (int)((e >> 6) % A.nk);
  const int t = (int)((e >> 6) / A.nk);
  const int m = lane & 15;
  const int i = 4 * s + (lane >> 4);
  int q, j, b;
  if (A.tpl > 0) {
    q = 0;
    j = t / A.tpl;
    b = (t % A.tpl) * 16 + m;
  } else {
    q = 4 * t + (m & 3);
    j = q % A.kp;
    b = 4 * (q / A.kp) + (m >> 2);
  }
  double val = 0.0;
  if (q < A.nquads && j < A.k && b < A.R && i < A.n) {
    const double *y = B.Yz + ((int64_t)b * A.n + i) * B.nb;
    const double *u = B.U + (int64_t)B.rowcell[i] * B.nb * A.k + j;
    for (int h = 0; h < B.nb; ++h) val = fma(y[h], u[(int64_t)h * A.k], val);
  }
  A.frag[e] = val;
}

// Index-select operators, one workgroup per tile: the index vectors and the
// columns of M that the tile's 16 batch columns need are staged in LDS once, so
// the n-long scan per fragment element reads LDS instead of global memory (same
// summation order as ops_kernel<false>, hence bit-identical fragments).
// LDS: 16 x n ints + 16 x n doubles.
__global__ __launch_bounds__(256) void ops_select_kernel(OpsArgs A) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double *Mc = smem;                                   // [16][n]
  int32_t *Is = (int32_t *)(smem + 16 * A.n);          // [16][n]
  const int t = blockIdx.x;
  const int tid = threadIdx.x;
  for (int e = tid; e < 16 * A.n; e += 256) {
    const int m = e / A.n, r = e - m * A.n;
    int q, j, b;
    if (A.tpl > 0) {
      q = 0;
      j = t / A.tpl;
      b = (t % A.tpl) * 16 + m;
    } else {
      q = 4 * t + (m & 3);
      j = q % A.kp;
      b = 4 * (q / A.kp) + (m >> 2);
    }
    const bool live = q < A.nquads && j < A.k && b < A.R;
    Is[e] = live ? A.inds[(int64_t)b * A.n + r] : -1;
    Mc[e] = live ? A.M[(int64_t)r * A.k + j] : 0.0;
  }
  __syncthreads();
  for (int o = tid; o < A.nk * 64; o += 256) {
    const int lane = o & 63, s = o >> 6;
    const int m = lane & 15;
    const int i = 4 * s + (lane >> 4);
    const int32_t *ib = Is + m * A.n;
    const double *mc = Mc + m * A.n;
    double val = 0.0;
    for (int r = 0; r < A.n; ++r)
      if (ib[r] == i) val += mc[r];
    A.frag[(int64_t)t * A.nk * 64 + o] = val;
  }
}

// ---------------------------------------------------------------------------
// slab reductions (deterministic, two-level)
// ---------------------------------------------------------------------------
// out[c][e] = sum_{s in chunk c} in[s][e]
__global__ __launch_bounds__(256) void slab_sum_kernel(const double *in, double *out, int64_t E,
                                                       int nslab, int chunk) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  const int c = blockIdx.y;
  const int s0 = c * chunk;
  const int s1 = min(nslab, s0 + chunk);
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  int s = s0;
  for (; s + 3 < s1; s += 4) {
    a0 += in[(int64_t)s * E + e];
    a1 += in[(int64_t)(s + 1) * E + e];
    a2 += in[(int64_t)(s + 2) * E + e];
    a3 += in[(int64_t)(s + 3) * E + e];
  }
  for (; s < s1; ++s) a0 += in[(int64_t)s * E + e];
  out[(int64_t)c * E + e] = (a0 + a1) + (a2 + a3);
}

// final level: sums nslab slabs of [C][w] and scatters tile-ordered columns to
// resample-major [R][k][w]
__global__ __launch_bounds__(256) void slab_final_kernel(const double *in, double *out, int64_t C,
                                                         int w, int nslab, int kp, int k, int R,
                                                         int nquads, int tpl) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= C * w) return;
  const int64_t c = e / w;
  const int cc = (int)(e % w);
  const int m = (int)(c & 15);
  int64_t q, b;
  int j;
  if (tpl > 0) {
    q = 0;
    j = (int)((c >> 4) / tpl);
    b = ((c >> 4) % tpl) * 16 + m;
  } else {
    q = 4 * (c >> 4) + (m & 3);
    j = (int)(q % kp);
    b = 4 * (q / kp) + (m >> 2);
  }
  if (q >= nquads || j >= k || b >= R) return;
  // four loads in flight (a single dependent chain over ~50 slabs is latency-bound)
  const int64_t stride = C * w;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  int s = 0;
  for (; s + 3 < nslab; s += 4) {
    a0 += in[(int64_t)s * stride + e];
    a1 += in[(int64_t)(s + 1) * stride + e];
    a2 += in[(int64_t)(s + 2) * stride + e];
    a3 += in[(int64_t)(s + 3) * stride + e];
  }
  for (; s < nslab; ++s) a0 += in[(int64_t)s * stride + e];
  out[(b * k + j) * w + cc] = (a0 + a1) + (a2 + a3);
}

// S[e] += sum over column splits of part[c][e]  (fixed order)
__global__ __launch_bounds__(256) void moment_merge_kernel(double *S, const double *part, int64_t count,
                                                           int nsplit) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= count) return;
  double a = part[e];
  for (int c = 1; c < nsplit; ++c) a += part[(int64_t)c * count + e];
  S[e] += a;
}

// std / bootstrap-ratio from shifted moments
__global__ __launch_bounds__(256) void boot_finalize_kernel(const double *S1, const double *S2,
                                                            const double *num, int64_t count,
                                                            double invR, double *sd, double *ratio) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= count) return;
  const double m = S1[e] * invR;
  const double var = fma(-m, m, S2[e] * invR);
  const double s = sqrt(var > 0.0 ? var : 0.0);
  sd[e] = s;
  if (ratio != nullptr) ratio[e] = num[e] / s;
}

// out[r][c] = in[r][c] * scale[c]  (the observed V s from V and s: bootstrap_permutation.py:695, :701)
__global__ __launch_bounds__(256) void scale_cols_kernel(const double *in, const double *scale, int64_t count,
                                                         int cols, double *out) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= count) return;
  out[e] = in[e] * scale[e % cols];
}

}  // namespace plsr
