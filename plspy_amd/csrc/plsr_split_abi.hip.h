// extern "C" entry points of K2s (two-stage split-half Gram, plsr_split.hip.h).  See include/plsr.h.
#include "plsr_split.hip.h"

namespace {
struct SplitPlan {
  int inst;                       // index into the instance table below
  int NTB, NTO, NTT, CS, MC, NQ;  // the instance's (maximum) counts
  bool exact;
  int cs, nx, csub, NG;
  int64_t ntile;
  size_t lds;
  size_t o_roff, o_bfrag, o_wfrag, o_gp, bytes;
};

struct SplitInstance {
  int NTB, NTO, NTT, CS;
  bool exact;
};
// exact instances first: config 4 (mb, two groups x three conditions, bscan of two: eight behaviour cells, twelve
// task cells, twelve task rows), the behaviour PLS of config 3's data (twelve behaviour cells) and the multiblock
// PERMUTATION of the same data (whole samples: four behaviour and six task cells of twenty rows)
const SplitInstance kSplitInst[] = {{4, 6, 1, 3, true},  {6, 0, 0, 3, true},  {2, 3, 1, 5, true}, {5, 5, 1, 3, false},
                                    {6, 0, 0, 3, false}, {6, 0, 0, 5, false}, {2, 4, 1, 5, false}};
// the ROWS variant (plsr_split_rows): exact for config 6 (mb bootstrap: four behaviour cells and six task cells of
// twenty rows), then the guarded ones
const SplitInstance kRowsInst[] = {{2, 3, 1, 5, true}, {5, 5, 1, 3, false}, {6, 0, 0, 3, false}, {6, 0, 0, 5, false},
                                   {2, 4, 1, 5, false}};

bool split_plan(int32_t n, int64_t ldx, int64_t p, int32_t b, const int32_t *cell_rows, int32_t nq, int32_t nbq,
                int32_t ktask, int32_t m, int32_t items, SplitPlan &pl, bool rows = false) {
  if (n <= 0 || p < 16 || ldx < p || b <= 0 || b > SG_BP || !cell_rows || nq <= 0 || nq > SG_MAXQ || nbq < 0 ||
      nbq > nq || ktask < 0 || ktask > 16 || m <= 0 || items <= 0)
    return false;
  if ((int64_t)n * ldx * 8 >= ((int64_t)1 << 32)) return false;      // 32-bit row offsets
  if (ktask == 0 && nbq != nq) return false;                         // task-only cells without task rows
  int cs = 1, rmin = 1 << 30;
  for (int q = 0; q < nq; ++q) {
    if (cell_rows[q] < 0) return false;
    cs = std::max(cs, (cell_rows[q] + 3) / 4);
    rmin = std::min(rmin, cell_rows[q]);
  }
  const int ntb = (nbq + SG_IP - 1) / SG_IP, np = (nq + SG_IP - 1) / SG_IP;
  pl.inst = -1;
  const SplitInstance *table = rows ? kRowsInst : kSplitInst;
  const int ntable = rows ? (int)(sizeof(kRowsInst) / sizeof(kRowsInst[0])) : (int)(sizeof(kSplitInst) / sizeof(kSplitInst[0]));
  for (int i = 0; i < ntable; ++i) {
    const SplitInstance &I = table[i];
    bool ok;
    if (I.exact)
      ok = nbq == I.NTB * SG_IP && nq == (I.NTB + I.NTO) * SG_IP && (ktask > 0) == (I.NTT > 0) && cs == I.CS &&
           rmin > 4 * (I.CS - 1);
    else
      ok = ntb <= I.NTB && np <= I.NTB + I.NTO && (ktask == 0 || I.NTT > 0) && cs <= I.CS;
    if (ok) {
      pl.inst = i;
      pl.NTB = I.NTB, pl.NTO = I.NTO, pl.NTT = I.NTT, pl.CS = I.CS, pl.exact = I.exact;
      break;
    }
  }
  if (pl.inst < 0) return false;
  pl.cs = cs;
  pl.MC = pl.NTB + pl.NTT;
  pl.NQ = (pl.NTB + pl.NTO) * SG_IP;
  pl.NG = pl.MC * (pl.MC + 1) / 2;
  if (m > pl.MC * 16) return false;
  pl.ntile = (p + 15) / 16;
  pl.nx = (int)std::min<int64_t>(8, pl.ntile);
  {
    // voxel ranges per item: the eight XCDs' eighths of X, cut further only while the launch has fewer than
    // about four rounds of the chip's 1024 one-wave slots -- and never below 64 tiles per wave (the row
    // offsets, the fragments and the 30 KB partial Gram are per wave)
    const int64_t per_x = (pl.ntile + pl.nx - 1) / pl.nx;
    const int64_t want = (4096 + (int64_t)items * pl.nx - 1) / ((int64_t)items * pl.nx);
    pl.csub = (int)std::max<int64_t>(1, std::min<int64_t>(want, per_x / 64));
  }
  pl.lds = split_lds_bytes(nbq, cs, pl.NQ * pl.CS, pl.exact, rows ? pl.MC : 0);
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t o = off;
    off += (bytes + 255) / 256 * 256;
    return o;
  };
  pl.o_roff = take((size_t)items * nq * cs * 4 * sizeof(uint32_t));
  pl.o_bfrag = take((size_t)items * std::max(nbq, 1) * cs * 64 * sizeof(double));
  pl.o_wfrag = take((size_t)((pl.NQ + 3) / 4) * 64 * sizeof(double));
  pl.o_gp = rows ? take((size_t)items * pl.nx * pl.csub * pl.MC * 16 * sizeof(double))
                 : take((size_t)items * pl.nx * pl.csub * pl.NG * 256 * sizeof(double));
  pl.bytes = off;
  return true;
}

template <int NTB, int NTO, int NTT, int CS, bool EXACT, bool ROWS = false>
int launch_split(const SplitArgs &a, const SplitPlan &pl, hipStream_t st) {
  auto kern = split_gram_kernel<NTB, NTO, NTT, CS, EXACT, ROWS>;
  hipLaunchKernelGGL(kern, dim3((unsigned)((int64_t)a.items * pl.nx * pl.csub)), dim3(64), pl.lds, st, a);
  return launch_ok();
}
}  // namespace

extern "C" size_t plsr_split_gram_workspace_bytes(int32_t n, int64_t ldx, int64_t p, int32_t b,
                                                  const int32_t *cell_rows, int32_t nq, int32_t nbq, int32_t ktask,
                                                  int32_t m, int32_t items) {
  SplitPlan pl;
  return split_plan(n, ldx, p, b, cell_rows, nq, nbq, ktask, m, items, pl) ? pl.bytes : 0;
}

namespace {
// plsr_split_gram (d_G) and plsr_split_rows (d_R): meta kernels, the main kernel, the reduction
int split_run(const double *d_X, int64_t ldx, int64_t p, int32_t n, const int32_t *d_xsrc, const int32_t *d_ysrc,
              int32_t nz, const double *d_Y, int32_t b, const int32_t *cell_rows, int32_t nq, int32_t nbq,
              const double *d_Wc, int32_t ktask, const int32_t *row_cell, const int32_t *row_sub, int32_t m,
              int32_t items, void *d_work, size_t work_bytes, void *stream, double *d_G, double *d_rownorm,
              int32_t normalise, double *d_R, int64_t ldv, double *d_rowsq, int64_t rowsq_stride) {
  const bool rows = d_R != nullptr;
  if (!d_X || !d_xsrc || !d_work || !cell_rows || !row_cell || !row_sub) return PLSR_EINVAL;
  if (nbq > 0 && (!d_ysrc || !d_Y)) return PLSR_EINVAL;
  if (ktask > 0 && !d_Wc) return PLSR_EINVAL;
  SplitPlan pl;
  if (!split_plan(n, ldx, p, b, cell_rows, nq, nbq, ktask, m, items, pl, rows)) return PLSR_EUNSUPPORTED;
  if (pl.bytes > work_bytes) return PLSR_EWORKSPACE;
  {
    int tot = 0;
    for (int q = 0; q < nq; ++q) tot += cell_rows[q];
    if (tot != nz) return PLSR_EINVAL;
  }
  hipStream_t st = (hipStream_t)stream;
  char *w = (char *)d_work;

  SplitMetaArgs ma;
  ma.xsrc = d_xsrc;
  ma.ysrc = d_ysrc;
  ma.Y = d_Y;
  ma.b = b;
  ma.items = items;
  ma.nz = nz;
  ma.nq = nq;
  ma.nbq = nbq;
  ma.cs = pl.cs;
  ma.cell_lo[0] = 0;
  for (int q = 0; q < nq; ++q) ma.cell_lo[q + 1] = ma.cell_lo[q] + cell_rows[q];
  ma.ldx_bytes = ldx * 8;
  ma.roff = (uint32_t *)(w + pl.o_roff);
  ma.bfrag = (double *)(w + pl.o_bfrag);
  {
    const int64_t total = (int64_t)items * nq * pl.cs * 4 + (int64_t)items * nbq * pl.cs * 64;
    hipLaunchKernelGGL(split_meta_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, ma);
  }
  double *wfrag = (double *)(w + pl.o_wfrag);
  if (pl.NTT) {
    if (ktask > 0) {
      hipLaunchKernelGGL(split_wfrag_kernel, dim3((unsigned)((pl.NQ + 3) / 4)), dim3(64), 0, st, d_Wc, ktask, nq, wfrag);
    } else if (hipMemsetAsync(wfrag, 0, (size_t)((pl.NQ + 3) / 4) * 64 * sizeof(double), st) != hipSuccess) {
      return PLSR_ELAUNCH;
    }
  }

  // logical row -> the kernel's row (tile * 16 + column), and back
  SplitReduceArgs ra;
  SplitArgs a;
  for (int i = 0; i < 7 * 16; ++i) a.rowof[i] = -1;
  for (int l = 0; l < m; ++l) {
    int at;
    if (row_cell[l] >= 0) {
      if (row_cell[l] >= nbq || row_sub[l] < 0 || row_sub[l] >= b) return PLSR_EINVAL;
      at = (row_cell[l] / SG_IP) * 16 + (row_cell[l] % SG_IP) * SG_BP + row_sub[l];
    } else {
      if (!pl.NTT || row_sub[l] < 0 || row_sub[l] >= ktask) return PLSR_EINVAL;
      at = (pl.MC - 1) * 16 + row_sub[l];
    }
    ra.inv[l] = (int16_t)at;
    a.rowof[at] = (int16_t)l;
  }

  a.X = d_X;
  a.p = p;
  a.items = items;
  a.nq = nq;
  a.nbq = nbq;
  a.cs = pl.cs;
  for (int q = 0; q < SG_MAXQ; ++q) {
    a.nrow[q] = q < nq ? cell_rows[q] : 0;
    a.rn[q] = a.nrow[q] > 0 ? 1.0 / (double)a.nrow[q] : 0.0;
  }
  a.roff = ma.roff;
  a.bfrag = ma.bfrag;
  a.wfrag = pl.NTT ? wfrag : nullptr;
  a.nx = pl.nx;
  a.csub = pl.csub;
  a.ntile = pl.ntile;
  a.Gp = rows ? nullptr : (double *)(w + pl.o_gp);
  a.R = d_R;
  a.ldv = ldv;
  a.m = m;
  a.rowsq_part = rows ? (double *)(w + pl.o_gp) : nullptr;
  int rc = PLSR_EUNSUPPORTED;
  if (!rows) {
    switch (pl.inst) {
      case 0: rc = launch_split<4, 6, 1, 3, true>(a, pl, st); break;
      case 1: rc = launch_split<6, 0, 0, 3, true>(a, pl, st); break;
      case 2: rc = launch_split<2, 3, 1, 5, true>(a, pl, st); break;
      case 3: rc = launch_split<5, 5, 1, 3, false>(a, pl, st); break;
      case 4: rc = launch_split<6, 0, 0, 3, false>(a, pl, st); break;
      case 5: rc = launch_split<6, 0, 0, 5, false>(a, pl, st); break;
      case 6: rc = launch_split<2, 4, 1, 5, false>(a, pl, st); break;
    }
  } else {
    switch (pl.inst) {
      case 0: rc = launch_split<2, 3, 1, 5, true, true>(a, pl, st); break;
      case 1: rc = launch_split<5, 5, 1, 3, false, true>(a, pl, st); break;
      case 2: rc = launch_split<6, 0, 0, 3, false, true>(a, pl, st); break;
      case 3: rc = launch_split<6, 0, 0, 5, false, true>(a, pl, st); break;
      case 4: rc = launch_split<2, 4, 1, 5, false, true>(a, pl, st); break;
    }
  }
  if (rc) return rc;

  ra.items = items;
  ra.nchunk = pl.nx * pl.csub;
  ra.MC = pl.MC;
  ra.m = m;
  ra.mm = (m + 15) / 16 * 16;
  if (rows) {
    hipLaunchKernelGGL(split_rowsq_kernel, dim3((unsigned)(((int64_t)items * m + 255) / 256)), dim3(256), 0, st,
                       (const double *)a.rowsq_part, items, ra.nchunk, pl.MC, m, rowsq_stride, ra, d_rowsq);
    return launch_ok();
  }
  ra.Gp = a.Gp;
  ra.normalise = normalise;
  ra.G = d_G;
  ra.rownorm = d_rownorm;
  const size_t lds = (size_t)(pl.MC * 16) * (pl.MC * 16 + 1) * sizeof(double);
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute((const void *)split_reduce_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
          hipSuccess)
    return PLSR_ELAUNCH;
  hipLaunchKernelGGL(split_reduce_kernel, dim3((unsigned)items), dim3(256), lds, st, ra);
  return launch_ok();
}
}  // namespace

extern "C" int plsr_split_gram(const double *d_X, int64_t ldx, int64_t p, int32_t n, const int32_t *d_xsrc,
                               const int32_t *d_ysrc, int32_t nz, const double *d_Y, int32_t b,
                               const int32_t *cell_rows, int32_t nq, int32_t nbq, const double *d_Wc, int32_t ktask,
                               const int32_t *row_cell, const int32_t *row_sub, int32_t m, int32_t normalise,
                               int32_t items, double *d_G, double *d_rownorm, void *d_work, size_t work_bytes,
                               void *stream) {
  if (!d_G) return PLSR_EINVAL;
  return split_run(d_X, ldx, p, n, d_xsrc, d_ysrc, nz, d_Y, b, cell_rows, nq, nbq, d_Wc, ktask, row_cell, row_sub, m,
                   items, d_work, work_bytes, stream, d_G, d_rownorm, normalise, nullptr, 0, nullptr, 0);
}

extern "C" size_t plsr_split_rows_workspace_bytes(int32_t n, int64_t ldx, int64_t p, int32_t b,
                                                  const int32_t *cell_rows, int32_t nq, int32_t nbq, int32_t ktask,
                                                  int32_t m, int32_t items) {
  SplitPlan pl;
  return split_plan(n, ldx, p, b, cell_rows, nq, nbq, ktask, m, items, pl, true) ? pl.bytes : 0;
}

extern "C" int plsr_split_rows(const double *d_X, int64_t ldx, int64_t p, int32_t n, const int32_t *d_xsrc,
                               const int32_t *d_ysrc, int32_t nz, const double *d_Y, int32_t b,
                               const int32_t *cell_rows, int32_t nq, int32_t nbq, const double *d_Wc, int32_t ktask,
                               const int32_t *row_cell, const int32_t *row_sub, int32_t m, int32_t items, double *d_R,
                               int64_t ldv, double *d_rowsq, int64_t rowsq_stride, void *d_work, size_t work_bytes,
                               void *stream) {
  if (!d_R || !d_rowsq || ldv < p || rowsq_stride < m) return PLSR_EINVAL;
  return split_run(d_X, ldx, p, n, d_xsrc, d_ysrc, nz, d_Y, b, cell_rows, nq, nbq, d_Wc, ktask, row_cell, row_sub, m,
                   items, d_work, work_bytes, stream, nullptr, nullptr, 0, d_R, ldv, d_rowsq, rowsq_stride);
}

// ---------------------------------------------------------------------------
// K4m: projection of the multiblock bootstrap from the stored products of the un-normalised rows
// ---------------------------------------------------------------------------
#include "plsr_rowsproj.hip.h"

namespace {
struct RowsProjPlan {
  int MC, ks, KSMAX, nsplit, per;
  size_t o_rdinv, o_ufrag, o_mom, bytes;
};

bool rows_project_plan(int32_t kr, int32_t k, int32_t items, int64_t p, bool moments, RowsProjPlan &pl) {
  if (kr <= 0 || k <= 0 || k > kr || kr > 48 || items <= 0 || p <= 0) return false;
  if ((int64_t)kr * p * 8 >= ((int64_t)1 << 32)) return false;          // 32-bit row offsets inside an item
  pl.MC = (k + 15) / 16;
  pl.ks = (kr + 3) / 4;
  pl.KSMAX = pl.MC == 1 && pl.ks <= 4 ? 4 : (pl.MC <= 2 && pl.ks <= 8 ? 8 : (pl.ks <= 10 ? 10 : 12));
  const int64_t nwg = (p + 63) / 64;
  // splits of the items: only where few voxel tiles leave the chip empty (every split writes its own moments)
  pl.nsplit = (int)std::max<int64_t>(1, std::min<int64_t>((1024 + nwg - 1) / nwg, std::max(1, items / 8)));
  pl.per = (items + pl.nsplit - 1) / pl.nsplit;
  pl.nsplit = (items + pl.per - 1) / pl.per;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t o = off;
    off += (bytes + 255) / 256 * 256;
    return o;
  };
  pl.o_rdinv = take((size_t)items * 4 * pl.ks * sizeof(double));
  pl.o_ufrag = take((size_t)pl.MC * pl.ks * 64 * sizeof(double));
  pl.o_mom = take(moments ? (size_t)2 * pl.nsplit * p * k * sizeof(double) : 0);
  pl.bytes = off;
  return true;
}
}  // namespace

extern "C" size_t plsr_rows_project_workspace_bytes(int32_t kr, int32_t k, int32_t items, int64_t p,
                                                    int32_t want_moments) {
  RowsProjPlan pl;
  return rows_project_plan(kr, k, items, p, want_moments != 0, pl) ? pl.bytes : 0;
}

extern "C" int plsr_rows_project(double *d_R, int64_t ldv, int64_t p, int32_t items, int32_t kr, const double *d_rowsq,
                                 int64_t rowsq_stride, const double *d_U, int32_t k, const double *d_ref, double *d_S1,
                                 double *d_S2, double *d_out, void *d_work, size_t work_bytes, void *stream) {
  if (!d_R || !d_rowsq || !d_U || !d_work || ldv < p || rowsq_stride < kr) return PLSR_EINVAL;
  if ((d_S1 == nullptr) != (d_S2 == nullptr)) return PLSR_EINVAL;
  RowsProjPlan pl;
  if (!rows_project_plan(kr, k, items, p, d_S1 != nullptr, pl)) return PLSR_EUNSUPPORTED;
  if ((int64_t)kr * ldv * 8 >= ((int64_t)1 << 32)) return PLSR_EUNSUPPORTED;       // (a stride wider than p)
  if (pl.bytes > work_bytes) return PLSR_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  char *w = (char *)d_work;
  RowsProjArgs a;
  a.R = d_R;
  a.out = d_out;
  a.ldv = ldv;
  a.p = p;
  a.items = items;
  a.kr = kr;
  a.k = k;
  a.ks = pl.ks;
  a.per = pl.per;
  a.rdinv = (double *)(w + pl.o_rdinv);
  a.ufrag = (double *)(w + pl.o_ufrag);
  a.S1 = d_S1 ? (double *)(w + pl.o_mom) : nullptr;
  a.S2 = d_S1 ? a.S1 + (size_t)pl.nsplit * p * k : nullptr;
  {
    const int64_t total = (int64_t)items * 4 * pl.ks + (int64_t)pl.MC * pl.ks * 64;
    hipLaunchKernelGGL(rows_project_meta_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d_rowsq,
                       rowsq_stride, d_U, items, kr, k, pl.ks, pl.MC, (double *)a.rdinv, (double *)a.ufrag);
  }
  const dim3 grid((unsigned)((p + 63) / 64), (unsigned)pl.nsplit);
#define PLSR_RP(M, K) \
  if (pl.MC == M && pl.KSMAX == K) hipLaunchKernelGGL((rows_project_kernel<M, K>), grid, dim3(256), 0, st, a);
  PLSR_RP(1, 4) PLSR_RP(1, 8) PLSR_RP(2, 8) PLSR_RP(1, 10) PLSR_RP(2, 10) PLSR_RP(3, 10) PLSR_RP(1, 12) PLSR_RP(2, 12)
  PLSR_RP(3, 12)
#undef PLSR_RP
  if (launch_ok() != PLSR_OK) return PLSR_ELAUNCH;
  if (d_S1) {
    const int64_t cnt = p * k;
    hipLaunchKernelGGL(moment_unshift_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, d_S1, d_S2,
                       (const double *)a.S1, (const double *)a.S2, d_ref, cnt, pl.nsplit, (double)items);
  }
  return launch_ok();
}
