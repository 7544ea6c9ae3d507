/*
 * plsr.h -- C ABI of the MI355X (gfx950) PLS resampling engine.
 *
 * plspy itself has no FFI: its hot path is NumPy called from three Python
 * seams.  Each entry point below names the reference
 * arithmetic it replaces (file:line under plspy/core of McIntosh-Lab/plspy).
 * A maintainer of the reference binds these with ctypes.
 *
 * Conventions
 *   - every pointer named d_* is a DEVICE pointer (hipMalloc / torch tensor
 *     .data_ptr()); the caller owns every buffer, the library allocates nothing.
 *   - `stream` is a hipStream_t passed as void* (0 = default stream).  All
 *     entry points only enqueue work; none synchronises.
 *   - return value: 0 = ok, <0 = error (plsr_strerror).  No C++ exception
 *     crosses the boundary.
 *   - fp64 everywhere, int32 indices, row-major, voxel = unit stride.
 */
#ifndef PLSR_H
#define PLSR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PLSR_OK 0
#define PLSR_EINVAL (-1)      /* bad argument / shape                          */
#define PLSR_EUNSUPPORTED (-2)/* shape outside what the kernels are built for  */
#define PLSR_EWORKSPACE (-3)  /* workspace too small                           */
#define PLSR_ELAUNCH (-4)     /* hip launch error (see plsr_last_hip_error)    */

#define PLSR_VOXEL_TILE 64    /* voxels owned by one workgroup                 */

int plsr_abi_version(void);
const char *plsr_strerror(int code);
int plsr_last_hip_error(void);

/*
 * Layout of a batch of R resamples x k latent variables as MFMA operand
 * fragments.  Column (b, j) of the batch is the n-vector `op` such that the
 * resample's projected cross-block is  VS_b[v, j] = sum_i X[i, v] * op[i].
 * Columns are grouped in "quads" (one latent variable, four consecutive
 * resamples) and four quads make one 16-row MFMA tile, so that the four
 * accumulator registers of a lane are four resamples of ONE latent variable
 * and the sum over resamples needs no cross-lane traffic.
 */
typedef struct plsr_layout {
  int32_t n;        /* rows of X                                             */
  int32_t k;        /* latent variables per resample                         */
  int32_t R;        /* resamples in the batch                                */
  int32_t nk;       /* k-steps: ceil(n / 4)                                  */
  int32_t kp;       /* k padded so that the tile pattern has a short period  */
  int32_t period;   /* tiles after which the (slot, lane-group) -> lv map repeats */
  int32_t Rp;       /* R rounded up to a multiple of 4                       */
  int32_t ntiles;   /* 16-column tiles in the batch                          */
  int64_t frag_elems; /* doubles in the fragment buffer: ntiles * nk * 64    */
} plsr_layout_t;

/* Fills *out.  PLSR_EUNSUPPORTED when k needs a period the kernels lack. */
int plsr_layout_init(int32_t n, int32_t k, int32_t R, plsr_layout_t *out);

/*
 * Operator columns from row-selection indices.
 *   op_(b,j)[i] = sum over r with inds[b][r] == i of M[r][j]      (r ascending)
 * With M = W^T U (W = the mean-centring operator, U = observed left singular
 * vectors) this is the fold of
 *     X[inds,:]                      resample.py:79, :153
 *     _mean_centre(...)              class_functions.py:7-95
 *     permuted.T @ U                 bootstrap_permutation.py:404, :620
 * into one (n x k) operator per resample, so X is never gathered.
 *   d_inds : [R][n] int32      d_M : [n][k] fp64      d_frag : frag_elems fp64
 */
int plsr_ops_from_indices(const int32_t *d_inds, const double *d_M,
                          const plsr_layout_t *lay, double *d_frag, void *stream);

/*
 * Operator columns given densely: d_cols[b][j][i] (R x k x n fp64), for
 * preprocessors whose operator is not a row selection (behaviour PLS).
 */
int plsr_ops_pack(const double *d_cols, const plsr_layout_t *lay, double *d_frag,
                  void *stream);
/*
 * Behaviour-PLS operators from the per-cell z-scored behaviour block of every
 * resample (class_functions.py:240-242 folded with `@ U`, as used at
 * bootstrap_permutation.py:404):
 *   Op_b[i][j] = sum_beh Yz[b][i][beh] * U[cell(i)*nb + beh][j]
 * d_Yz [R][n][nb], d_U [ncell*nb][k], d_rowcell [n] int32 (cell of every row);
 * the R x k x n operator columns are never formed on the host.
 */
int plsr_ops_from_behaviour(const double *d_Yz, int32_t nb, const double *d_U,
                            const int32_t *d_rowcell, const plsr_layout_t *lay, double *d_frag,
                            void *stream);

/* Bytes of scratch the two batch calls below need for p voxels. */
size_t plsr_batch_workspace_bytes(const plsr_layout_t *lay, int64_t p, int32_t k2);

/*
 * Launch shape the two batch calls below choose for (lay, p, k2) -- a query for tests and
 * profiles, no device work.  out[0] = voxel tiles (grid.x); out[1] = batch-column splits
 * (LV-major layout: runs per latent variable, grid.y = k * out[1]); out[2] = batch tiles per
 * latent variable (LV-major) or total batch tiles (quad layout); out[3] = 1 if the
 * register-resident kernels (n <= 64) serve, else 0.  `boot` selects the bootstrap launch.
 * (No reference counterpart: the reference runs one resample per loop iteration,
 * bootstrap_permutation.py:323, :537.)
 */
int plsr_batch_plan(const plsr_layout_t *lay, int64_t p, int32_t k2, int32_t boot, int32_t out[4]);

/*
 * Permutation batch.  For every resample b and latent variable j:
 *     d_ssq[b][j] = sum_v ( sum_i X[i,v] * op_(b,j)[i] )^2
 * i.e. s_hat^2 of bootstrap_permutation.py:404-405.
 *   d_X : [n][ldx] fp64 (ldx >= p)        d_ssq : [R][k] fp64 (overwritten)
 */
int plsr_perm_batch(const double *d_X, int64_t ldx, int64_t p,
                    const double *d_frag, const plsr_layout_t *lay,
                    double *d_ssq, void *d_work, size_t work_bytes, void *stream);

/*
 * Bootstrap batch.  With VS_b = X^T op_b (p x k), streams instead of storing
 * right_sv_sampled (bootstrap_permutation.py:497, :626, :695):
 *     d_S1[v][j] += sum_b (VS_b[v,j] - ref[v][j])
 *     d_S2[v][j] += sum_b (VS_b[v,j] - ref[v][j])^2         (shifted moments)
 *     d_ssq[b][j] = sum_v VS_b[v,j]^2                        (:623 norms)
 *     d_T[b][j][c] = sum_v Xm[c][v] * VS_b[v,j]              (:633-634 before
 *                    the column normalisation, with Xm = the k2 x p cell means
 *                    of X: group_condition_means(X @ V_hat) = (Wm X) V_hat)
 *   d_ref : [p][k] or NULL (= 0)      d_Xm : [k2][ldxm], k2 <= 16, or NULL
 *   d_vs_dump : [R][p][k] or NULL     (materialised VS, for tests / debug)
 * d_S1/d_S2 are accumulated into (caller zeroes them before the first batch).
 */
int plsr_boot_batch(const double *d_X, int64_t ldx, int64_t p,
                    const double *d_frag, const plsr_layout_t *lay,
                    const double *d_ref, const double *d_Xm, int64_t ldxm, int32_t k2,
                    double *d_S1, double *d_S2, double *d_ssq, double *d_T,
                    double *d_vs_dump,
                    void *d_work, size_t work_bytes, void *stream);

/*
 * out[r][c] = in[r][c] * scale[c] for a row-major (rows x cols) matrix: the observed V s
 * (the shift of the bootstrap moments and the numerator of boot_ratios,
 * bootstrap_permutation.py:695-703) from V and s where only those are at hand.  In place allowed.
 */
int plsr_scale_cols(const double *d_in, int64_t rows, int32_t cols, const double *d_scale,
                    double *d_out, void *stream);

/*
 * Bootstrap summary from the streamed moments (bootstrap_permutation.py:695-703):
 *     std[e]   = sqrt( max( S2[e]/R - (S1[e]/R)^2 , 0 ) )      np.std, ddof = 0
 *     ratio[e] = num[e] / std[e]                               boot_ratios
 * for e in [0, count); S1/S2 are the shifted sums of plsr_boot_batch summed over
 * all batches (and all GPUs).  d_num = V*s (or V with contrasts).
 */
int plsr_boot_finalize(const double *d_S1, const double *d_S2, const double *d_num,
                       int64_t count, int32_t R, double *d_std, double *d_ratio, void *stream);

/*
 * ---- K2: cross-block Gram and thin SVD -------------------------------------
 * An item is one decomposition request: m operator rows A (m x n) whose
 * cross-block M = A X (m x p) is never stored.  Replaces _run_pls ->
 * np.linalg.svd (class_functions.py:98-123) as used by
 * split_half_resampling.py:194-196, :612-613, :682-683 and pls_classes.py:261:
 * with G = M M^T = U S^2 U^T,  s = sqrt(eig), V = M^T U / s.
 */
/* doubles in the fragment buffer for `items` items of m rows */
int64_t plsr_rows_frag_elems(int32_t n, int32_t m, int32_t items);
/* d_rows[item][j][i] (items x m x n fp64) -> MFMA operand fragments */
int plsr_ops_pack_rows(const double *d_rows, int32_t items, int32_t m, int32_t n,
                       double *d_frag, void *stream);
size_t plsr_gram_workspace_bytes(int32_t n, int32_t m, int32_t items, int64_t p,
                                 int64_t x_item_stride);
/* d_G[item][mm][mm], mm = 16*ceil(m/16); rows/cols >= m are zero.
 * x_item_stride = 0: all items contract the same X; otherwise item i uses the
 * n x ldx matrix at d_X + i*x_item_stride (per-item z-scored data, K3). */
int plsr_gram_batch(const double *d_X, int64_t x_item_stride, int64_t ldx, int64_t p, int32_t n,
                    const double *d_frag, int32_t items, int32_t m, double *d_G,
                    void *d_work, size_t work_bytes, void *stream);
/*
 * Symmetric eigen-decomposition of the k x k block at (off, off) of each of
 * `count` matrices (leading dimension ld, item_stride doubles apart): one
 * wavefront per matrix, cyclic Jacobi in LDS.  Eigenvalues descending in
 * d_evals[item][k]; eigenvectors in the columns of d_evecs[item][k][k].
 * k <= 64.
 */
int plsr_eigh_batch(const double *d_G, int64_t item_stride, int32_t ld, int32_t off,
                    int32_t k, int32_t count, double *d_evals, double *d_evecs,
                    const double *d_init, int32_t relative, void *stream);
/*
 * (continued) d_init: NULL, or [count][k][k] -- the rotations are then applied to the columns
 * of d_init[item] instead of the identity, so d_evecs = d_init * J: the basis accumulated over
 * refinement passes (must not alias d_evecs).  relative != 0: the block is the Gram of rows
 * that are already nearly orthogonal with graded norms (a refinement pass); no rotation is
 * skipped for being small against the largest diagonal entry, which makes the eigenvalues
 * RELATIVELY accurate -- LAPACK-grade small singular values for _run_pls's np.linalg.svd
 * (class_functions.py:98-123), which eig of a once-formed Gram cannot give (it squares the
 * condition number).
 *
 * plsr_rotate_rows: d_rows_out[item][off + j][:] = sum_i d_U[item][i][j] * d_rows_in[item][off + i][:]
 * for j < k; rows outside [off, off + k) are copied.  The operator rows of a decomposed block in
 * its eigenvector basis: input of the next refinement pass's Gram, and after the last pass the
 * operator of the back-projection V s = (rows_out @ X)^T.
 */
int plsr_rotate_rows(const double *d_U, const double *d_rows_in, double *d_rows_out, int32_t items,
                     int32_t m, int32_t n, int32_t off, int32_t k, void *stream);
/*
 * Last step of the thin SVD, on the device (no host round trip between the decomposition and the
 * resampling phases): d_lam [k] eigenvalues of the final pass (descending), d_cur [k][n] = U^T rows.
 *   d_s[i]            = sqrt(lam_i), or 0 for a null latent variable: s_i <= max(abs_tol, rel_tol * s_0)
 *                       (abs_tol = the reference's 1e-12, bootstrap_permutation.py:295)
 *   d_rows_out[i]     = cur[i]        (0 if null):  rows_out[:k] @ X = (V s)^T   (class_functions.py:122-123)
 *   d_rows_out[k + i] = cur[i] / s_i  (0 if null):  rows_out[k:] @ X =  V^T
 */
int plsr_svd_finish(const double *d_lam, const double *d_cur, int32_t k, int32_t n, double abs_tol,
                    double rel_tol, double *d_s, double *d_rows_out, void *stream);

/*
 * ---- K3: row gather + per-cell z-score --------------------------------------
 * The data side of _compute_corr (class_functions.py:185-247): for every item,
 *   out[item][r][:] = X[src[item][r]][:]                         for rows of raw cells
 *   out[item][r][:] = zscore over the cell's rows (ddof 0) / sqrt(n_cell)   for z cells
 * cells are the output-row ranges [cell_lo[c], cell_lo[c+1]).  Voxels that are
 * constant within a cell give 0 there (scipy's zscore rule + nan_to_num).
 * Contracting the result with z-scored behaviour columns gives R = Yz.T @ Xz.
 *   d_src : [items][nout] int32     d_out : [items][nout][ldo] fp64
 */
int plsr_gather_zscore(const double *d_X, int64_t ldx, int64_t p, const int32_t *d_src,
                       int32_t items, int32_t nout, const int32_t *d_cell_lo,
                       const int32_t *d_cell_z, int32_t ncell, double *d_out, int64_t ldo,
                       void *stream);

/*
 * ---- K5: latent scores of the behaviour / multiblock bootstrap -----------------
 * plsr_latent:  d_Zt[b][j][i] = sum_v VS_b[j][v] X[i][v]   (X @ VS_b,
 *     bootstrap_permutation.py:638/:647/:655 before the column normalisation)
 *     d_nsq[b][j]   = sum_v VS_b[j][v]^2                       (:623 norms; d_nsq may be
 *     null -- plsr_item_fused already returns these norms as its row norms)
 * from d_vst[b][j][v] = VS_b[j][v] (k x ldv per item, written by plsr_item_fused).
 * k <= 64, n <= 256.
 */
size_t plsr_latent_workspace_bytes(int32_t n, int32_t k, int32_t items, int64_t p);
int plsr_latent(const double *d_X, int64_t ldx, int64_t p, int32_t n, const double *d_vst,
                int64_t ldv, int32_t items, int32_t k, double *d_Zt, double *d_nsq, void *d_work,
                size_t work_bytes, void *stream);

/*
 * K5x: plsr_latent with X read as pre-transposed B fragments (n <= 128).  d_XT is X^T, voxel-major
 * with rows padded to 128 doubles and voxels padded with zero rows to whole 32-voxel tiles
 * (plsr_latent_xt_bytes), made once per X by plsr_latent_xt_prepare.  Every wave loads its own B
 * fragments (four voxels x its sixteen data rows = four 128-byte segments) straight into
 * registers a tile ahead; only VS^T goes through LDS (double-buffered, one barrier per tile).
 * Same outputs and the same limits on k as plsr_latent.
 */
size_t plsr_latent_xt_bytes(int32_t n, int64_t p);
int plsr_latent_xt_prepare(const double *d_X, int64_t ldx, int64_t p, int32_t n, double *d_XT, void *stream);
size_t plsr_latent_xt_workspace_bytes(int32_t n, int32_t k, int32_t items, int64_t p);
int plsr_latent_xt(const double *d_XT, int64_t p, int32_t n, const double *d_vst, int64_t ldv, int32_t items,
                   int32_t k, double *d_Zt, double *d_nsq, void *d_work, size_t work_bytes, void *stream);

/*
 * K5i: d_L[b][j][i] = sum_v VS_b[j][v] X[d_idx[b][i]][v], i < m -- the latent scores of the SAMPLE,
 * `_compute_X_latents(X_new, V_hat)` with X_new = X[inds] before the column normalisation
 * (bootstrap_permutation.py:638, :646; class_functions.py:165-182).  A bootstrap sample of n rows holds
 * about 0.63 n different rows: a small kernel lists them per item (ascending); a wave owns sixteen
 * entries of the list and reads those rows from d_XB, a tile-major copy of X ([tile of 32 voxels][n]
 * [32], voxels past p zero; plsr_latent_xb_bytes / plsr_latent_xb_prepare, once per X): one 32-byte
 * load per lane = one 128-byte line per row and instruction, VS^T goes through LDS as in K5x; waves
 * past an item's list only stage.  The chunk sum scatters the columns back to the sample's order.
 * n <= 128, k <= 64.
 * max_rows: the caller's bound on the number of different rows per item (<= n; it sets the workgroup
 * size, one wave per sixteen); an item that exceeds it, or holds an index outside [0, n), comes back
 * as NaN.  d_nsq as for plsr_latent.  vst_tiled != 0: d_vst is tile-major (see plsr_item_beh).
 * d_T non-null: t_rows (<= 16) further columns d_L[b][j][m + t] = sum_v VS_b[j][v] T_b[d_t_row[t]][v] with
 * T_b = rows of the item's own block d_T[b] ([t_item_rows][t_ld], row-major) -- the multiblock bootstrap's
 * Tdistrib: the cell means of smeanmat(X_new_T) @ V_hat are the sample's raw task rows times V_hat
 * (bootstrap_permutation.py:652-656; class_functions.py:479-490), and those rows are what plsr_split_rows
 * left in R_b, so the latent kernel needs the rows of the BEHAVIOUR sample only.  d_L is [items][k][m + t_rows].
 */
size_t plsr_latent_xb_bytes(int32_t n, int64_t p);
int plsr_latent_xb_prepare(const double *d_X, int64_t ldx, int64_t p, int32_t n, double *d_XB, void *stream);
size_t plsr_latent_index_workspace_bytes(int32_t n, int32_t k, int32_t items, int64_t p, int32_t m,
                                         int32_t max_rows, int32_t t_rows);
int plsr_latent_index(const double *d_XB, int64_t p, int32_t n, const double *d_vst, int64_t ldv,
                      int32_t vst_tiled, int32_t items, int32_t k, const int32_t *d_idx, int32_t m,
                      int32_t max_rows, const double *d_T, int64_t t_ld, int32_t t_item_rows,
                      const int32_t *d_t_row, int32_t t_rows, double *d_L, double *d_nsq, void *d_work,
                      size_t work_bytes, void *stream);

/*
 * ---- K0: a handful of operator rows applied to X ------------------------------
 * d_out (m x p, row stride ldo) = d_rows (m x n, row-major) @ X.  The observed
 * blocks of a PLS() call: _mean_centre / cell means as the operator W
 * (class_functions.py:7-95, pls_classes.py:211-266), the behaviour correlation
 * block on the z-scored X (:185-247), the multiblock (:454-516), contrast
 * projections (:126-162) and the back-projection V = M^T U / s of _run_pls
 * (:98-123).  HBM-bound (X is read once per 16 output rows).  n <= 512.
 */
int plsr_apply_rows(const double *d_X, int64_t ldx, int64_t p, int32_t n, const double *d_rows,
                    int32_t m, double *d_out, int64_t ldo, void *stream);

/*
 * ---- K4f: fused gather / z-score / projection ---------------------------------
 * Every bootstrap sample of behaviour / multiblock PLS z-scores its own
 * resampled rows (class_functions.py:221-238 on X[inds]), so each resample
 * ("item") has its own matrix Z_b and its own k x nz operator rows;
 * VS_b = rows_b Z_b.  Same result as plsr_gather_zscore followed by a dense
 * product, without the per-item matrix ever reaching HBM: a workgroup keeps X[:, 64 voxels] in LDS
 * for all items, gathers rows through d_src, z-scores them on the fly from
 * per-(item, cell, voxel) statistics (two-pass, computed by a first kernel from
 * the same LDS tile) and multiplies by the item's operator rows.
 *   d_src   : [items][nz] int32 source row of every row of the item matrix
 *   cell_lo : HOST [ncell+1] row ranges of the cells, cell_z : HOST [ncell]
 *             (1 = z-score the cell as class_functions.py:221-238, 0 = copy rows);
 *             ncell <= 64
 *   d_rows  : [items][k][nz] operator rows (VS_b = rows_b @ Z_b)
 *   d_S1/d_S2 (both or neither): [p][k] += shifted moment sums (shift d_ref [p][k] or NULL)
 *   d_vst   : [items][k][ldv] VS^T or NULL
 *   d_rowsq : [items][ceil(k/16)*16]  sum_v VS_b[j][v]^2  or NULL
 *             (bootstrap_permutation.py:623 norms; class_functions.py:503-505 row norms)
 *   d_sc / d_sh : NULL (the per-(item, cell, voxel) scale and shift live in the workspace), or
 *             caller buffers [items][ncell][p]: filled by this call when stats_ready == 0,
 *             taken as they are when stats_ready != 0 -- the multiblock bootstrap calls twice
 *             on the same items (row norms, then projection) and computes them once
 * k <= 128, n <= 320.
 */
size_t plsr_item_fused_workspace_bytes(int32_t n, int32_t nz, int32_t k, const int32_t *cell_lo,
                                       int32_t ncell, int32_t items, int64_t p, int32_t want_moments,
                                       int32_t want_rowsq);
int plsr_item_fused(const double *d_X, int64_t ldx, int64_t p, int32_t n, const int32_t *d_src,
                    int32_t nz, const int32_t *cell_lo, const int32_t *cell_z, int32_t ncell,
                    const double *d_rows, int32_t items, int32_t k, const double *d_ref, double *d_S1,
                    double *d_S2, double *d_vst, int64_t ldv, double *d_rowsq, double *d_sc,
                    double *d_sh, int32_t stats_ready, void *d_work, size_t work_bytes, void *stream);

/*
 * ---- K4a: the same product on aggregated operators, X in registers -------------
 * plsr_item_fused for items whose cells draw their rows from FIXED RANGES of source rows --
 * what the reference's bootstrap does (resample.py:132-160: subjects within their group, the
 * same draw for every condition; bootstrap_permutation.py:547-557), so cell c of every item
 * reads rows src_lo[c] <= r < src_hi[c] of X only.  The gather then folds into the operator
 * (A_bc[j][r] = sum of rows_b[j][i] over the cell's rows i with d_src[b][i] == r), every
 * k-step of every item reads the same rows of X, and a wave keeps X[:, 16 voxels] in
 * registers for all items; the per-(item, cell, voxel) statistics of four items at a time
 * come from v_mfma_f64_4x4x4 on the same registers (no statistics kernel, nothing through HBM).
 * Replaces class_functions.py:185-247 + :454-516 on X[inds] and the projection
 * bootstrap_permutation.py:620 inside the bootstrap loop (:537-675) for rb / mb / csb / cmb.
 *   src_lo / src_hi : HOST [ncell] source-row range of every cell.  An item that reads a row
 *             outside its cell's range comes out as NaN (all its outputs and the moment sums).
 *   other arguments and outputs as plsr_item_fused (no caller-held statistics).
 * Differences in arithmetic from plsr_item_fused: the variance is formed in one pass on data
 * centred by the per-voxel grand mean; a cell whose sample variance is below 16 eps of its
 * second moment about that mean is treated as constant (z = 0), next to the reference's
 * sd <= eps |mean| rule.
 * plsr_item_agg_workspace_bytes returns 0 when the shape is not served (n > 128, k > 48,
 * more than 16 z-scored cells, or cells whose ranges overlap so much that the dense sweeps
 * would cost over 4/3 of plsr_item_fused's k-steps): call plsr_item_fused then.
 */
size_t plsr_item_agg_workspace_bytes(int32_t n, int32_t nz, int32_t k, const int32_t *cell_lo,
                                     const int32_t *cell_z, const int32_t *src_lo, const int32_t *src_hi,
                                     int32_t ncell, int32_t items, int64_t p, int32_t want_moments,
                                     int32_t want_rowsq);
int plsr_item_agg(const double *d_X, int64_t ldx, int64_t p, int32_t n, const int32_t *d_src, int32_t nz,
                  const int32_t *cell_lo, const int32_t *cell_z, const int32_t *src_lo,
                  const int32_t *src_hi, int32_t ncell, const double *d_rows, int32_t items, int32_t k,
                  const double *d_ref, double *d_S1, double *d_S2, double *d_vst, int64_t ldv,
                  double *d_rowsq, void *d_work, size_t work_bytes, void *stream);

/*
 * ---- K4b: behaviour PLS bootstrap in two stages ---------------------------------
 * The same VS_b as plsr_item_agg with operator rows  rows_b[j][i] = sum_beh Yz_b[i][beh] *
 * U[(cell(i), beh)][j]  (behaviour PLS: class_functions.py:185-247 followed by
 * bootstrap_permutation.py:620), taken in the reference's own two steps instead of one folded
 * product: R_bc = Yz_bc^T Z_bc per cell (b rows), then VS_b = sum_c U_c^T R_bc -- 2 b n p +
 * 2 k^2 p flops per item instead of 2 k n p (config 3: 1.3 instead of 2.3 GFLOP).  Two items
 * share the 16 rows of a stage-1 MFMA tile (b <= 8); the scaled stage-1 accumulators are, in
 * place, the B operands of stage 2.  All cells are z-scored.
 *   d_Yz : [items][nz][b] behaviour rows of every item's sample, z-scored within the cells
 *          (their columns must sum to zero over a cell: the shift of X's z-score then drops out)
 *   d_U  : [ncell * b][k] left singular vectors, rows ordered (cell, behaviour)
 *   cell_lo / src_lo / src_hi : HOST arrays as in plsr_item_agg (an item that reads outside its
 *          ranges comes out as NaN); d_S1 / d_S2 / d_ref / d_vst as in plsr_item_fused;
 *          vst_tiled != 0: d_vst is written TILE-MAJOR, [items][ldv / 32][k][32] (ldv a multiple of 32:
 *          a 32-voxel tile's k rows of 256 bytes side by side) -- the layout plsr_latent_index streams
 *          (a workgroup's 64 voxels leave as two contiguous blocks of k x 256 bytes per item instead of
 *          k pieces of 512 bytes 8 ldv apart, and a wave of the latent kernel reads its voxel range as one
 *          contiguous block)
 * b <= 16, k <= 48, ncell <= 16, and (cells x k-steps of the longest cell) must fit one of the
 * register layouts (5 x 6, 8 x 4, 4 x 8, 2 x 16): plsr_item_beh_workspace_bytes returns 0 otherwise.
 */
size_t plsr_item_beh_workspace_bytes(int32_t n, int32_t nz, int32_t b, int32_t k, const int32_t *cell_lo,
                                     const int32_t *src_lo, const int32_t *src_hi, int32_t ncell,
                                     int32_t items, int64_t p, int32_t want_moments);
int plsr_item_beh(const double *d_X, int64_t ldx, int64_t p, int32_t n, const int32_t *d_src, int32_t nz,
                  const int32_t *cell_lo, const int32_t *src_lo, const int32_t *src_hi, int32_t ncell,
                  const double *d_Yz, int32_t b, const double *d_U, int32_t items, int32_t k,
                  const double *d_ref, double *d_S1, double *d_S2, double *d_vst, int64_t ldv, int32_t vst_tiled,
                  void *d_work, size_t work_bytes, void *stream);

/*
 * Multiblock operator rows for plsr_item_fused, formed on the device from the
 * un-normalised rows and the squared row norms plsr_item_fused returned for them
 * (two-phase row normalisation, class_functions.py:503-505, then `@ U`, :620):
 *   d_out[b][j][i] = sum_r d_U[r][j] / sqrt(d_rowsq[b][r]) * d_raw[b][r][i]
 * (a row of norm 0 contributes 0).  d_raw [items][kr][nz], d_rowsq [items][rowsq_stride],
 * d_U [kr][k], d_out [items][k][nz].
 */
int plsr_scale_project_rows(const double *d_raw, const double *d_rowsq, int64_t rowsq_stride,
                            const double *d_U, int32_t items, int32_t kr, int32_t nz, int32_t k,
                            double *d_out, void *stream);

/*
 * plsr_gram_batch with the gather / z-score of K4f fused in (split-half of
 * behaviour / multiblock PLS, split_half_resampling.py:136-398): item b's matrix
 * is X[d_src[b]] z-scored within the cells, never stored; G_b = (rows_b Z_b)(rows_b Z_b)^T.
 *   d_src : [items][nz] int32, cell_lo / cell_z : HOST arrays as in plsr_item_fused,
 *   d_frag: plsr_ops_pack_rows(items, m, nz)
 * nz <= 256 (the kernel keeps an item's source-row and cell tables in registers), m <= 96.
 */
size_t plsr_gram_fused_workspace_bytes(int32_t n, int32_t nz, int32_t m, const int32_t *cell_lo,
                                       int32_t ncell, int32_t items, int64_t p);
int plsr_gram_fused(const double *d_X, int64_t ldx, int64_t p, int32_t n, const int32_t *d_src,
                    int32_t nz, const int32_t *cell_lo, const int32_t *cell_z, int32_t ncell,
                    const double *d_frag, int32_t items, int32_t m, double *d_G, void *d_work,
                    size_t work_bytes, void *stream);

/*
 * K2s: the per-split Grams of behaviour / multiblock PLS in two stages (plsr_split.hip.h), the fast path of
 * the split-half tests (split_half_resampling.py:119-197, :266-383, :548-683, :687-802; the cross-blocks of
 * class_functions.py:185-247 and :454-516 are never stored).  An item's rows of X are grouped in CELL SLOTS
 * (the half's group x condition cells): the first nbq slots carry b behaviour rows each (the cell's rows of
 * X z-scored per voxel, times the cell's rows of Y z-scored per column), the others only feed the task rows,
 * which are linear in the cells' sums:  T[j] = sum_q d_Wc[j][q] * (sum of the rows of slot q).
 *   d_xsrc, d_ysrc : [items][nz] int32, rows of X / of d_Y per (slot, row), slots concatenated (d_ysrc is
 *                    read for the first nbq slots only)
 *   cell_rows      : HOST [nq] rows per slot (sum = nz)
 *   d_Y [.][b], b <= 8;  d_Wc [ktask][nq] (ktask <= 16) or null
 *   row_cell / row_sub : HOST [m], what row l of the item's stacked cross-block is: behaviour row row_sub[l]
 *                    of slot row_cell[l] >= 0, or (row_cell[l] < 0) task row row_sub[l]
 *   normalise      : G_ij / (sqrt(G_ii) sqrt(G_jj)) -- the multiblock row normalisation
 *                    (class_functions.py:503-505) applied to the Gram; rows of norm 0 give 0
 *   d_G            : [items][mm][mm], mm = 16 ceil(m / 16), rows / columns past m are 0
 *   d_rownorm      : [items][mm] or null: the norms over all voxels of the un-normalised rows
 * The workspace query returns 0 for shapes the kernel's instances do not serve (b > 8, cells of more than 12
 * / 20 rows, more than 10-16 cells, X of 4 GiB or more, p < 16): plsr_gram_fused serves those.
 */
size_t plsr_split_gram_workspace_bytes(int32_t n, int64_t ldx, int64_t p, int32_t b, const int32_t *cell_rows,
                                       int32_t nq, int32_t nbq, int32_t ktask, int32_t m, int32_t items);
int plsr_split_gram(const double *d_X, int64_t ldx, int64_t p, int32_t n, const int32_t *d_xsrc,
                    const int32_t *d_ysrc, int32_t nz, const double *d_Y, int32_t b, const int32_t *cell_rows,
                    int32_t nq, int32_t nbq, const double *d_Wc, int32_t ktask, const int32_t *row_cell,
                    const int32_t *row_sub, int32_t m, int32_t normalise, int32_t items, double *d_G,
                    double *d_rownorm, void *d_work, size_t work_bytes, void *stream);

/*
 * The ROWS variant of K2s: the same cell description, but the item's (un-normalised) cross-block rows themselves
 * are the result -- d_R [items][m][ldv] in logical row order -- with their squared norms over all voxels,
 * d_rowsq [items][rowsq_stride].  First pass of the multiblock bootstrap (bootstrap_permutation.py:547-553, :610;
 * class_functions.py:454-516 without the row normalisation): the rows of X a bootstrap sample draws are read by
 * index (repeats and all), per cell; plsr_rows_project then normalises, projects and accumulates the moments.
 */
size_t plsr_split_rows_workspace_bytes(int32_t n, int64_t ldx, int64_t p, int32_t b, const int32_t *cell_rows,
                                       int32_t nq, int32_t nbq, int32_t ktask, int32_t m, int32_t items);
int plsr_split_rows(const double *d_X, int64_t ldx, int64_t p, int32_t n, const int32_t *d_xsrc,
                    const int32_t *d_ysrc, int32_t nz, const double *d_Y, int32_t b, const int32_t *cell_rows,
                    int32_t nq, int32_t nbq, const double *d_Wc, int32_t ktask, const int32_t *row_cell,
                    const int32_t *row_sub, int32_t m, int32_t items, double *d_R, int64_t ldv, double *d_rowsq,
                    int64_t rowsq_stride, void *d_work, size_t work_bytes, void *stream);

/*
 * K4m: the second pass of the multiblock bootstrap as a stream over the first pass's products
 * (class_functions.py:503-505, bootstrap_permutation.py:610, :620, :695).  plsr_item_agg / plsr_item_fused, run
 * with the UN-NORMALISED multiblock rows as operator, leave R_b = raw_b Z_b (kr x p) in d_R and the squared norms
 * of its rows in d_rowsq; this call replaces R_b by
 *     VS_b^T[j][v] = sum_r d_U[r][j] / sqrt(d_rowsq[b][r]) * R_b[r][v]        (rows 0 .. k - 1 of item b's block)
 * in place (a row of norm 0 contributes 0) and adds the shifted moment sums of VS_b over the items to d_S1 / d_S2
 * ([p][k], shift d_ref or none), as plsr_item_agg does.  d_R [items][kr][ldv], d_U [kr][k], k <= kr <= 48,
 * kr * ldv * 8 < 4 GiB (the query returns 0 otherwise; assumes ldv = p).  d_out non-null: VS_b^T goes to
 * d_out [items][k][ldv] instead and R_b stays (plsr_latent_index reads the raw task rows from it).
 */
size_t plsr_rows_project_workspace_bytes(int32_t kr, int32_t k, int32_t items, int64_t p, int32_t want_moments);
int plsr_rows_project(double *d_R, int64_t ldv, int64_t p, int32_t items, int32_t kr, const double *d_rowsq,
                      int64_t rowsq_stride, const double *d_U, int32_t k, const double *d_ref, double *d_S1,
                      double *d_S2, double *d_out, void *d_work, size_t work_bytes, void *stream);

/*
 * ---- F4: the upstream feed, X built on the device ---------------------------------
 * plspy/io/io.py:427-460 (apply_mask_matrices: `m[np.broadcast_to(mask, m.shape)]` -- for every
 * time point the voxels the mask selects, in C order) and :680-698 (concat_flatten_all_groups:
 * all subjects stacked, one row each), without the masked copies on the host:
 *   plsr_mask_indices    stream compaction of the mask: d_idx[j] = flat index of the j-th selected
 *                        voxel (ascending), *d_count = their number (device int64).  d_mask: nvox bytes.
 *   plsr_mask_apply_rows d_out[r][j] = d_in[r][d_idx[j]] for nrows rows (time points of a subject's
 *                        volume, row stride ld_in) into rows of X (row stride ld_out, fp64); the
 *                        source is fp64 or fp32 (in_is_f32).  nrows <= 65535.
 * HBM-bound byte work; the reference's own property for these two functions is the round trip of
 * plspy/tests/test_io.py:8-36 (tests/test_gpu_io.py).
 */
size_t plsr_mask_indices_workspace_bytes(int64_t nvox);
int plsr_mask_indices(const uint8_t *d_mask, int64_t nvox, int64_t *d_idx, int64_t *d_count, void *d_work,
                      size_t work_bytes, void *stream);
int plsr_mask_apply_rows(const void *d_in, int32_t in_is_f32, int64_t ld_in, int64_t nrows, const int64_t *d_idx,
                         int64_t nsel, double *d_out, int64_t ld_out, void *stream);

/*
 * ---- host: bit-exact NumPy legacy RandomState draws -------------------------
 * (no GPU involved; these run wherever the library loads).  key[624] / *pos are
 * np.random.get_state()[1:3]; they come back advanced so that
 * np.random.set_state() continues the reference's stream.  Replace the Python
 * loops around np.random.permutation / np.random.choice in
 * resample.py:44-77, :125-160 and split_half_resampling.py:136,271,282,316.
 *   table : [nsub][nc] int32 row ids (subjects x conditions, groups stacked)
 */
/* `count` draws of np.random.permutation(n) -> out[count][n] */
int plsr_rng_permutation(uint32_t *key, int32_t *pos, int32_t n, int32_t count, int32_t *out);
/* `count` rounds of np.random.permutation(sizes[0]), ..., np.random.permutation(sizes[nsizes - 1]) in that
 * order -> out[count][sum(sizes)]: the per-group subject shuffles of a split (split_half_resampling.py:136) and
 * the subject + row shuffles of a null split (:271, :282 / :316) */
int plsr_rng_permutation_seq(uint32_t *key, int32_t *pos, const int32_t *sizes, int32_t nsizes, int32_t count,
                             int32_t *out);
/* `count` task-PLS permutations (resample.py:63-73) -> out[count][nsub*nc] */
int plsr_rng_task_permutations(uint32_t *key, int32_t *pos, const int32_t *table, int32_t nsub,
                               int32_t nc, int32_t count, int32_t *out);
/* `count` bootstrap samples (resample.py:132-160) -> out[count][nsub*nc];
 * group_subjects[g] = subjects in group g */
int plsr_rng_bootstraps(uint32_t *key, int32_t *pos, const int32_t *table,
                        const int32_t *group_subjects, int32_t ngroups, int32_t nc, int32_t count,
                        int32_t *out);
/* `count` tries of the multiblock permutation (bootstrap_permutation.py:343-347): per try a
 * task permutation, then np.random.permutation(nrows) for the behaviour block */
int plsr_rng_mb_permutations(uint32_t *key, int32_t *pos, const int32_t *table, int32_t nsub,
                             int32_t nc, int32_t nrows, int32_t count, int32_t *out_task,
                             int32_t *out_rows);
/* `count` tries of the multiblock bootstrap (:547-553): per try a task bootstrap on `table`
 * (nc conditions), then a behaviour bootstrap on `btable` (bnc conditions, same groups) */
int plsr_rng_mb_bootstraps(uint32_t *key, int32_t *pos, const int32_t *table,
                           const int32_t *group_subjects, int32_t ngroups, int32_t nc,
                           const int32_t *btable, int32_t bnc, int32_t count, int32_t *out_task,
                           int32_t *out_beh);

/*
 * Optional overlap of the reduction tail.  plsr_boot_batch ends with HBM-bound
 * kernels (moment merges, slab sums of the norm and T partials) that do not
 * need the matrix cores.  With a tail stream set (per calling thread; NULL
 * restores the default), plsr_boot_batch enqueues the projection kernel on
 * `stream` and the reductions on the tail stream, ordered after the projection
 * by an event, so that whatever the caller enqueues next on `stream` overlaps
 * them.  The outputs (d_S1, d_S2, d_ssq, d_T) and d_work are then owned by the
 * tail stream: the caller must make its stream wait for the tail stream before
 * touching them.  plsr_boot_batch itself makes `stream` wait for the tail stream
 * on entry (consecutive batches share a workspace).
 */
int plsr_set_tail_stream(void *stream);

/*
 * Kernel timing for the roofline report (bench.py).  When enabled, every
 * projection-kernel launch made by plsr_perm_batch / plsr_boot_batch is
 * bracketed by hipEvents recorded on the launch stream.  plsr_timing_collect
 * synchronises those events and returns the elapsed milliseconds of each launch
 * since the last collect (kind[i] = 0 perm, 1 boot), at most `max` of them.
 * Not for use inside a graph capture.
 */
int plsr_timing_enable(int on);
int plsr_timing_collect(double *ms_out, int32_t *kind_out, int32_t max);

#ifdef __cplusplus
}
#endif
#endif /* PLSR_H */
