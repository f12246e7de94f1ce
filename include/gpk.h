/*
 * gpk.h — C ABI of libgpk.so: MI355X (gfx950) kernels for the Gaussian-Process
 * residual-model path of Grandediw/Unmanned_Aerial_Vehicles.
 *
 * The reference has no FFI boundary: its GP arithmetic is Python calling
 * scikit-learn / SciPy (LAPACK) on the CPU.  Each entry point below replaces one
 * of those CPU call sites (cited per function as <file>:<line> relative to the
 * reference tree; `sklearn/` = scikit-learn 1.7.2, the library the reference
 * delegates to).  The Python host side (unmanned_aerial_vehicles_amd/) binds these
 * with ctypes; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - All matrices are row-major.  "dev" pointers are device (HBM) addresses, e.g.
 *     torch.Tensor.data_ptr() on ROCm; "host" pointers are ordinary host memory.
 *   - Dense factor matrices are padded: Np = gpk_padded(N) = N rounded up to 128.
 *     The Gram kernel fills the padding with the identity, so the Cholesky factor of
 *     the padded matrix is [L 0; 0 I] and every dense kernel runs on whole tiles.
 *   - dtype: GPK_F32 or GPK_F64 selects the element type of the void* buffers.
 *   - Calls are asynchronous on the handle's stream unless stated otherwise;
 *     gpk_synchronize() waits.  Functions with a host output parameter synchronise.
 *   - Return codes: GPK_OK, GPK_NOT_PD (Cholesky pivot <= 0; 1-based row in the
 *     error string and *info), GPK_BAD_ARG, GPK_HIP_ERROR.  gpk_last_error() gives text.
 *   - No global state; one handle per GPU / host thread.  Kernels launch on, and allocations come from, the calling
 *     thread's current HIP device: gpk_set_stream (and every allocation the handle makes) selects the handle's device,
 *     so a caller that drives handles on several GPUs from one thread calls gpk_set_stream before each group of calls.
 */
#ifndef GPK_H
#define GPK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gpk_context* gpk_handle;

/* the exported entry points (libgpk.so is built with -fvisibility=hidden: nothing else leaves the library) */
#define GPK_API __attribute__((visibility("default")))

enum { GPK_F32 = 0, GPK_F64 = 1 };
enum { GPK_OK = 0, GPK_NOT_PD = 1, GPK_BAD_ARG = 2, GPK_HIP_ERROR = 3 };
/* GPK_MAX_D: features accepted by the Gram kernels (gpk_gram, gpk_cross_gram_t and with them gpk_predict_var*).
 * GPK_MAX_D_PREDICT: features accepted by the fused mean, the one-call serving kernels and the gradient -
 * gpk_predict_mean*, gpk_predict_host*, gpk_lml_grad - and therefore by the composite gpk_fit / gpk_predict /
 * gpk_lml and by the Python host side (the reference's largest model has 16 inputs,
 * quadrotor_gp_mpc/quadrotor_gp_mpc/gaussian_process.py:66).                                                   */
enum { GPK_TILE = 128, GPK_MAX_D = 64, GPK_MAX_D_PREDICT = 16, GPK_MAX_P = 16, GPK_MAX_BATCH = 8 };

/* ---- context ------------------------------------------------------------------ */
GPK_API int gpk_create(gpk_handle* h, int device);
GPK_API void gpk_destroy(gpk_handle h);
GPK_API const char* gpk_last_error(gpk_handle h);
/* Launch on `stream` (a hipStream_t, e.g. torch.cuda.current_stream().cuda_stream; NULL is HIP's
 * default stream).  GPK_OWN_STREAM selects the non-blocking stream created by gpk_create, which is
 * what a fresh handle uses.                                                                      */
#define GPK_OWN_STREAM ((void*)(intptr_t)-1)
GPK_API int gpk_set_stream(gpk_handle h, void* stream);
GPK_API int gpk_synchronize(gpk_handle h);
GPK_API int64_t gpk_padded(int64_t n);

/* ---- measurement aid -----------------------------------------------------------------------------------
 * gpk_timing(h, 1): from now on the handle brackets its dominant launches with HIP events recorded on the
 * handle's stream - tag GPK_TIMED_K5: the one GEMM launch of gpk_predict_var_inv / gpk_predict_var_inv_split
 * (V = W K*^T with the column-norm epilogue); tag GPK_TIMED_GRAM: the Gram kernel of gpk_gram; tag GPK_TIMED_GRAD: the
 * streaming pass of gpk_lml_grad over K^-1; tag GPK_TIMED_POTRF: the launches of one gpk_potrf - and keeps the
 * last 64 pairs.  gpk_kernel_times synchronises the stream and returns the elapsed milliseconds of the bracketed
 * launches with that tag still in the ring, oldest first (*n_out of them, at most max_n).  bench.py uses it to
 * report the dominant kernel's duration over exactly the timed steps; rocprofv3's kernel trace of the same run is
 * the cross-check.  No reference counterpart (the reference has no instrumentation on this path).        */
/* gpk_set_option: the handle's tuning knobs (the library reads NOTHING from the environment), settable on a live handle:
 * "k5_split2_tile" (gpk_predict_var_inv_split2: 0 = the tallest of the 512 / 256 / 128 x 128 tiles that still comes in at
 * least 512 tiles, 1 = always 128 x 128, 2 = 512 x 128 whenever Np % 512 == 0), "k5_super", "small_path", "trsm256",
 * "trtri_levels", "gemm_small_tiles" / "gemm_tiny_tiles" (launches of fewer 128 x 128 tiles than these - 1024 / 320 - run on 64 x 64 /,
 * fp64 only, 32 x 32 tiles; bit-identical results), "k3_stream_min_np", "ptile" (gpk_potrf: 1 = the one-launch tile factorisation of
 * gpk_ptile.hip for 512 <= Np <= "ptile_max_np" (24576), 0 = the recursive launch chain), "ptile_prog_max_nt" (that launch:
 * up to this many tile columns (128 = always) the tiles under a diagonal tile follow its factorisation 16 columns at a time,
 * 0 = they wait for the whole inverse tile), "ptile_prog_rows" (1 .. 8 such tiles per column; 8), "ptile_single_max_nt" (up to this many tile
 * columns (96) the launch keeps one workgroup per CU instead of two, 0 = always two), "ptile_sr" / "ptile_sr_max_nt" (1: launches of up to
 * that many tile columns (36) run the 256-register build with two k-tiles in flight; bit-identical factors), "ptile_inv_max_np" (gpk_lml_eval
 * with a gradient: up to this padded size (4608) the tiles of the inverse factor are tasks of the same launch, 0 = always the
 * level-by-level products of gpk_trtri; same values to rounding), "ptile_xcd" / "ptile_xcd_min_nt" / "ptile_grp_rows" /
 * "ptile_grp_cols" (XCD-aware dealing of that launch's tasks: 0 = one global ticket counter (default), 1 = one queue per XCD with
 * the tile rows dealt round-robin, 2 = groups of rows x cols tiles per queue; bit-identical factors, measured slower:
 * profiles/r05_ptile_xcd_ab.log), "ptile_slots" (resident workgroups of that launch; 0 = by the rule above), "gemm_balanced"
 * (tile GEMMs whose tiles differ in k-range - the products with triangular operands of gpk_trtri / gpk_wtw / gpk_potrs_inv:
 * 1 = the balanced persistent tile schedule, 0 = the static tile mapping; bit-identical results), "gemm_balanced_max_tiles",
 * "gemm_wm_f64" / "gemm_wm_f32" (wave rows per GEMM workgroup, 2 or 4), "gemm_log" (1: every tile-GEMM launch to stderr),
 * "debug_fill" (1: the handle's scratch and serving work area are overwritten with NaN bytes at every request - the test
 * suite runs with it).  Used by the A/B timings and by the tests that pin a fast path to its plain form.
 * gpk_set_option_str: "ptile_trace_path" - the NEXT one-launch factorisation writes its per-task time stamps to that file
 * (debugging aid, tools/exp_ptile_trace.py).                                                                          */
GPK_API int gpk_set_option(gpk_handle h, const char* name, int value);
GPK_API int gpk_set_option_str(gpk_handle h, const char* name, const char* value);
enum { GPK_TIMED_K5 = 1, GPK_TIMED_GRAM = 2, GPK_TIMED_GRAD = 3, GPK_TIMED_POTRF = 4 };
GPK_API int gpk_timing(gpk_handle h, int enable);
GPK_API int gpk_kernel_times(gpk_handle h, int tag, double* ms, int max_n, int* n_out);

/* ---- batched mode: `count` (<= 8) same-shaped problems per call --------------------------------------
 * Between gpk_batch_begin and gpk_batch_end, gpk_potrf, gpk_leaf_inverses, gpk_trtri, gpk_wtw and
 * gpk_potrs_inv work on `count` independent problems with ONE launch chain: every kernel of the chain
 * gets a batch grid dimension.  The pointer arguments address problem 0; a pointer that lies inside a
 * buffer registered with gpk_batch_buffer(base, stride_bytes) advances by that stride per problem, any
 * other pointer is shared by all problems.  gpk_potrf's `info` then receives `count` entries.
 * (BASELINE configuration 5: the per-axis ax/ay/az GPs of src/px4/gp_trainer.py:139-179 factorised
 * together.)                                                                                          */
GPK_API int gpk_batch_begin(gpk_handle h, int count);
GPK_API int gpk_batch_buffer(gpk_handle h, const void* base, int64_t stride_bytes);
GPK_API int gpk_batch_end(gpk_handle h);
GPK_API const char* gpk_version(void);

/* ---- K1: RBF Gram build ---------------------------------------------------------
 * K[i][j] = sf2 * exp(-0.5 * sum_d ((x_id - x_jd) / ls_d)^2), exact differences,
 * diagonal = sf2 + diag_add, padding rows/cols (>= N) = identity.  Symmetric tiles are
 * computed once and written to both halves.
 * Replaces: sklearn/gaussian_process/kernels.py:1553-1560 (RBF.__call__, pdist + exp +
 * squareform), :1402 (WhiteKernel diag), sklearn/gaussian_process/_gpr.py:347 (alpha
 * jitter); quadrotor_gp_mpc/quadrotor_gp_mpc/gaussian_process.py:158-171.
 * X: dev (N x D), ls: host double[D], K: dev (Np x ldk), ldk >= Np.                  */
GPK_API int gpk_gram(gpk_handle h, int dtype, const void* X, int64_t N, int D, const double* ls,
             double sf2, double diag_add, void* K, int64_t ldk);

/* Row slab of the same matrix, for a Gram build sharded over GPUs by rows (no exchange: rank r writes the rows it
 * owns; the slabs only have to meet on one device if that device is to factorise): Kslab (dev, gpk_padded(nrows) x ldk)
 * receives rows row0 .. row0 + gpk_padded(nrows) - 1 of the padded matrix gpk_gram would write - every entry computed
 * directly (no symmetric mirroring across slabs), diagonal = sf2 + diag_add, identity in the padding.  row0 % 128 == 0.
 * Replaces the same reference lines as gpk_gram.                                                                    */
GPK_API int gpk_gram_rows(gpk_handle h, int dtype, const void* X, int64_t N, int D, const double* ls, double sf2,
                  double diag_add, int64_t row0, int64_t nrows, void* Kslab, int64_t ldk);

/* Cross kernel, transposed layout: B[j][m] = sf2 * exp(-0.5 ||(x_j - xq_m)/ls||^2) for
 * j < N, m < M; zero elsewhere in the (Np x Mp) padded block.  No white noise.
 * Replaces: sklearn/gaussian_process/kernels.py:1564-1565 (cdist + exp).
 * X dev (N x D), Xq dev (M x D), B dev (Np x ldb), ldb >= Mp = gpk_padded(M).          */
GPK_API int gpk_cross_gram_t(gpk_handle h, int dtype, const void* X, int64_t N, const void* Xq,
                     int64_t M, int D, const double* ls, double sf2, void* B, int64_t ldb);

/* ---- K2: blocked Cholesky --------------------------------------------------------
 * In-place lower Cholesky of the padded fp64 matrix A (Np x lda): recursive blocking,
 * 128x128 leaf factorisation in LDS, fp64-MFMA trsm/syrk/gemm tiles.  The strict upper
 * triangle is not referenced and is left as written by gpk_gram.  winv (dev, Np x 128)
 * receives the inverse of every 128x128 diagonal block of L (used by the solves).
 * *info (host) = 0, or the 1-based index of the first non-positive pivot (GPK_NOT_PD).
 * Synchronises.  Replaces: scipy.linalg.cholesky(K, lower=True) at
 * sklearn/gaussian_process/_gpr.py:349,587; gaussian_process.py:184.                  */
GPK_API int gpk_potrf(gpk_handle h, double* A, int64_t Np, int64_t lda, double* winv, int* info);

/* Recompute winv (inverses of the 128x128 diagonal blocks) from an existing padded factor L,
 * e.g. one imported from a scikit-learn pickle (L_ at sklearn/gaussian_process/_gpr.py:349). */
GPK_API int gpk_leaf_inverses(gpk_handle h, const double* L, int64_t Np, int64_t ldl, double* winv);

/* Convert the lower triangle of L and winv to fp32 copies (for the fp32 predict path). */
GPK_API int gpk_factor_to_f32(gpk_handle h, const double* L, int64_t Np, int64_t ldl, const double* winv,
                      float* Lf, int64_t ldlf, float* winvf);

/* ---- K3: alpha = L^-T (L^-1 Y) -----------------------------------------------------
 * Y: dev (N x P) row-major fp64 (already normalised), alpha: dev (N x P).  P <= GPK_MAX_P.
 * Replaces: cho_solve((L, True), y) at sklearn/gaussian_process/_gpr.py:360-364,597;
 * gaussian_process.py:187-189.                                                          */
GPK_API int gpk_potrs(gpk_handle h, const double* L, int64_t Np, int64_t ldl, const double* winv,
              const double* Y, int64_t N, int P, double* alpha);

/* K3 through the explicit inverse factor W = L^-1 (gpk_trtri): alpha = W^T (W Y), two GEMM launches
 * that each stream W once, instead of the 4 Np/128 - 2 launches of the recursive solve.              */
GPK_API int gpk_potrs_inv(gpk_handle h, const double* W, int64_t Np, int64_t ldw, const double* Y, int64_t N,
                  int P, double* alpha);

/* ---- K5 building blocks: B <- L^-1 B, and column sums of squares ---------------------
 * B: dev (Np x ldb) with Mp = multiple of 128 columns in use.  dtype selects fp32/fp64
 * (L, winv and B must all have that dtype).
 * Replaces: solve_triangular(L_, K_trans.T, lower=True) at sklearn/_gpr.py:454-456 and
 * the einsum at :477.                                                                    */
GPK_API int gpk_trsm_lower_left(gpk_handle h, int dtype, const void* L, int64_t Np, int64_t ldl,
                        const void* winv, void* B, int64_t Mp, int64_t ldb);
/* out[m] = sum_{i < Np} B[i][m]^2, accumulated in fp64; out: dev double[Mp].              */
GPK_API int gpk_colsumsq(gpk_handle h, int dtype, const void* B, int64_t Np, int64_t Mp, int64_t ldb,
                 double* out);

/* ---- K4: fused posterior mean -----------------------------------------------------------
 * mean[m][p] = y_mean[p] + y_std[p] * sum_j k(xq_m, x_j) alpha[j][p]; K* is never stored.
 * X dev (N x D), alpha dev (N x P), Xq dev (M x D), mean dev (M x P), all of `dtype`;
 * ls, y_mean, y_std: host double arrays.  D <= GPK_MAX_D_PREDICT, P <= GPK_MAX_P (GPK_BAD_ARG otherwise).
 * Queries must be finite (the host side validates them as scikit-learn does; a NaN coordinate gives k* = 0).
 * Replaces: sklearn/gaussian_process/_gpr.py:441-447 (K_trans @ alpha_, undo normalisation);
 * the 25-call loop at src/px4/mpc.py:1490-1506; gaussian_process.py:223-226.              */
GPK_API int gpk_predict_mean(gpk_handle h, int dtype, const void* X, const void* alpha, int64_t N, int D,
                     int P, const double* ls, double sf2, const double* y_mean,
                     const double* y_std, const void* Xq, int64_t M, void* mean);

/* One-call serving for the control loop (fp64): host queries in, host posterior mean (and variance) out.
 * Copies Xq (host, M x D) through a pinned staging block owned by the handle, runs gpk_predict_mean and -
 * when var_host != NULL - gpk_predict_var_inv with the explicit inverse factor W (dev Np x ldw, gpk_trtri),
 * copies the results back and synchronises the handle's stream: one call and one synchronisation per MPC
 * step instead of an upload, two launch chains and two downloads driven from the host language.
 * mean_host: M x P (un-normalised with y_mean / y_std); var_host: M (normalised-target units: the caller
 * scales by y_std^2), clipped below at floor_.  1 <= M <= GPK_HOST_MAX_M.
 * Replaces: the per-call path of src/px4/simple_gp.py:187-201 (predict_residual) and the 25-call loop of
 * src/px4/mpc.py:1490-1506; sklearn/_gpr.py:441-494.
 * Up to 32 queries (D, P <= 16, N <= 16384) take two dedicated launches (K* + mean shares; 16 rows of W per
 * workgroup on the fp64 MFMA, last-workgroup reductions) instead of the general chain's seven.           */
#define GPK_HOST_MAX_M 4096
GPK_API int gpk_predict_host(gpk_handle h, const double* X, const double* alpha, int64_t N, int D, int P,
                     const double* ls, double sf2, const double* y_mean, const double* y_std,
                     const double* W, int64_t Np, int64_t ldw, double kss, double floor_,
                     const double* Xq_host, int64_t M, double* mean_host, double* var_host);

/* The same one-call serving for B <= 8 single-output models that share the query batch - the per-axis GPs of
 * gp_trainer.py / pretrained_gp.py - in ONE call and two launches (model = second grid dimension).
 * X, alpha, W: arrays of B device pointers (N x D, N x 1, Np x ldw); ls: B x D (host); sf2, y_mean, y_std, kss:
 * B (host).  1 <= M <= 32, D <= 16, N <= 16384.  mean_host: B x M (un-normalised); var_host: B x M in
 * normalised-target units, or NULL (then W and kss may be NULL).
 * Replaces: the loop over six scalar GPs of src/px4/pretrained_gp.py:52-98.                                 */
GPK_API int gpk_predict_host_multi(gpk_handle h, int B, const double* const* X, const double* const* alpha, int64_t N, int D,
                           const double* ls, const double* sf2, const double* y_mean, const double* y_std,
                           const double* const* W, int64_t Np, int64_t ldw, const double* kss, double floor_,
                           const double* Xq_host, int64_t M, double* mean_host, double* var_host);

/* K4 on the matrix cores, fp32 only: the same posterior mean as gpk_predict_mean(GPK_F32, ...), with the
 * pairwise squared distances of 32 x 32 (query, training point) blocks formed by MFMAs from centred, scaled
 * coordinates (|a|^2 + |b|^2 - 2 a.b as ONE augmented dot product of depth D + 2) and only exp2 + P FMAs per
 * pair left on the vector ALU.  D <= 14: six v_mfma_f32_32x32x16_bf16 per block on an exact three-way bf16
 * split of every fp32 operand (fp32-accurate distances); the training side is centred, scaled and split once
 * per call into MFMA-fragment order (N x 96 bytes of the handle's scratch) and loaded from L2 straight into
 * registers.  D = 15, 16: nine v_mfma_f32_32x32x2_f32.  The training set is cut into chunks of <= 2048
 * points whose partial sums are added in fp64.  center: host double[D], a point near the data (the
 * training mean); the expansion is accurate to ~|u|^2 * 2^-23 in the exponent, u = (x - center) / ls, so
 * callers use it while max |u|^2 is modest (device.py: <= 64) and the exact-difference kernel otherwise.
 * D <= 16, P <= 8.  Replaces the same reference lines as gpk_predict_mean.                              */
GPK_API int gpk_predict_mean_mfma(gpk_handle h, const float* X, const float* alpha, int64_t N, int D, int P,
                          const double* ls, double sf2, const double* center, const double* y_mean,
                          const double* y_std, const float* Xq, int64_t M, float* mean);

/* K5 on the bf16 matrix pipe at fp32 accuracy (exact operand split).
 * gpk_split3: src (dev rows x ld fp32, cols % 16 == 0) -> dst (dev, rows * cols * 6 bytes): every fp32 value
 * as three bf16 parts x = x0 + x1 + x2 (8 significant bits each, rounded to nearest: the sum is exact), stored as
 * 16-byte chunks [row / 4][k16 block][row % 4][half][part] (rows % 4 == 0; 96 bytes per row and 16 columns).
 * gpk_predict_var_inv_split: the same result as gpk_predict_var_inv(GPK_F32, ...) with W3 = gpk_split3 of the
 * fp32 inverse factor (gpk_tril_to_f32 output, Np x Np): one launch whose 32 x 32 x 16 block products are six
 * v_mfma_f32_32x32x16_bf16 each (a0 b0, a0 b1, a1 b0, a1 b1, a0 b2, a2 b0; exact bf16 products, fp32
 * accumulation; the dropped terms are below 2^-24 of |a||b|).  work: dev float[Mp * Np] (K* in fp32),
 * work3: dev, Mp * Np * 6 bytes (its split); var: dev double[Mp].  X, Xq fp32.
 * Replaces the same reference lines as gpk_predict_var (sklearn/gaussian_process/_gpr.py:454-485).       */
GPK_API int gpk_split3(gpk_handle h, const float* src, int64_t rows, int64_t cols, int64_t ld, void* dst);
GPK_API int gpk_predict_var_inv_split(gpk_handle h, const float* X, int64_t N, int D, const double* ls, double sf2,
                              const void* W3, int64_t Np, const float* Xq, int64_t M, double kss, double floor_,
                              float* work, void* work3, double* var);

/* The same launch with an fp16 x 2 operand split, three products per block, both operands in "fragment order" and one
 * scale per 128-row block of W (the fp32 serving default of the Python host side and of gpk_predict).
 * gpk_split2_rows: W (dev n x ld fp32 inverse factor, gpk_tril_to_f32 output, n % 128 == 0) -> scales (dev float[n / 128]:
 * for each 128-row block the largest power of two s with s * max |W_ij| <= 2^15, the maximum taken over the block's part of
 * the lower triangle - computed on the device, no synchronisation) and dst (dev, n * n * 4 bytes): x s = h0 + h1 + r with
 * h0, h1 fp16 rounded to nearest, |r| <= max(2^-23 |x s|, 2^-25) (relative for entries within 2^-17 of 2^15, absolute
 * below that, where h1 is a subnormal fp16); 16-byte chunks in fragment order: chunk (row, k16 block kb,
 * k half h, part p) at index (((row / 32) * (n / 16) + kb) * 2 + p) * 64 + h * 32 + row % 32 - the 64 chunks of one
 * v_mfma_f32_32x32x16_f16 operand are 1 KiB of contiguous memory.
 * gpk_predict_var_inv_split2: the result of gpk_predict_var_inv(GPK_F32, ...) from W2 / w_scales = gpk_split2_rows of the
 * fp32 inverse factor: K* (scaled by the power of two below 2^15 / sf2) is computed straight into the same layout
 * (work2: dev, Mp * Np * 4 bytes; no fp32 panel exists), and ONE launch forms V = W K*^T on the fp16 matrix pipe - block
 * products a1 b0 + a0 b1 + a0 b0 on v_mfma_f32_32x32x16_f16, fp32 accumulation: half the matrix-pipe work of the bf16 x 3
 * split - with every operand fragment loaded from L2 straight into registers (no LDS), reduced to column sums of squares
 * in its epilogue.  With round-to-nearest parts a0 + a1 reproduces a to 2^-23 at worst and the dropped a1 b1 is below
 * 2^-22 |a b|; measured over 31 random models the error of |W k*|^2 equals that of the exact-fp32 MFMA launch
 * (profiles/r02_fp32_variance_forms_accuracy.log).  var: dev double[Mp].  D <= 16.
 * Replaces the same reference lines as gpk_predict_var (sklearn/gaussian_process/_gpr.py:454-485).                  */
GPK_API int gpk_split2_rows(gpk_handle h, const float* W, int64_t n, int64_t ld, float* scales, void* dst);
/* gpk_split2_rows_f64: the same scales and parts straight from the fp64 inverse factor (gpk_trtri output, dev n x ld doubles):
 * every entry is rounded to fp32 on the fly exactly as gpk_tril_to_f32 would have stored it, so the result is bit-identical
 * on the lower tiles while the fp32 copy (17 GB at N = 65 536) and two passes over it are gone; the 16-column blocks right of
 * a row's diagonal tile - which the variance launch never reads - are left unwritten.                                    */
GPK_API int gpk_split2_rows_f64(gpk_handle h, const double* W, int64_t n, int64_t ld, float* scales, void* dst);
/* gpk_split2_rows_f64_absmax: the same with the first pass over W done already: on entry scales[n / 128] holds max |(float)W_ij|
 * over the lower triangle of each 128-row block - what gpk_trtri_absmax leaves behind - and the call turns it into the scales
 * and writes the parts: ONE pass over W (read 8, write 4 bytes per lower-tile entry) instead of two.  Same bits.             */
GPK_API int gpk_split2_rows_f64_absmax(gpk_handle h, const double* W, int64_t n, int64_t ld, float* scales, void* dst);
GPK_API int gpk_predict_var_inv_split2(gpk_handle h, const float* X, int64_t N, int D, const double* ls, double sf2,
                                       const void* W2, const float* w_scales, int64_t Np, const float* Xq, int64_t M,
                                       double kss, double floor_, void* work2, double* var);

/* One fp32 serving step in one call, the whole result in one buffer: out (dev double, M x 2P) row m =
 * [mean[m][0..P) | var[m] y_std[p]^2, p = 0..P) - K4 (gpk_predict_mean_mfma when `center` is given, else
 * gpk_predict_mean(GPK_F32)) into mean_tmp (dev float[M * P]), then gpk_predict_var_inv_split2's launch with the variance's
 * un-normalisation (sklearn/gaussian_process/_gpr.py:487-489) and the packing folded into its finalising kernel; that
 * kernel also adds to *low_count (dev, nullable; a running counter: the caller reads it before and after, or zeroes
 * it) the number of rows whose normalised variance is below `recheck_below` - the rows the fp32 serving gate
 * recomputes in fp64.
 * gpk_pack_mean_var: the same rows from a separate mean (dev M x P of `dtype`) and normalised variance (dev double[M]) -
 * for the serving paths whose variance comes from another launch.  P <= 16.                                          */
GPK_API int gpk_predict_mean_var_split2(gpk_handle h, const float* X, const float* alpha, int64_t N, int D, int P,
                                        const double* ls, double sf2, const double* center, const double* y_mean,
                                        const double* y_std, const void* W2, const float* w_scales, int64_t Np,
                                        const float* Xq, int64_t M, double kss, double floor_, void* work2,
                                        float* mean_tmp, double recheck_below, unsigned* low_count, double* out);
GPK_API int gpk_pack_mean_var(gpk_handle h, int dtype, const void* mean, const double* var, int64_t M, int P,
                              const double* y_std, double* out);

/* K4 for B (<= 8) independent single-output ARD models that share X (the per-axis GPs of
 * src/px4/gp_trainer.py:139-179, predicted one by one at src/px4/pretrained_gp.py:64-91): one launch
 * evaluates every model; the feature differences of a (query, training point) pair are formed once.
 * alpha: dev (N x B), column b = model b; ls: host double[B*D] (row b = model b's ARD length-scales);
 * sf2, y_mean, y_std: host double[B]; mean: dev (M x B).  X, alpha, Xq, mean of `dtype`.  D <= 16.      */
GPK_API int gpk_predict_mean_multi(gpk_handle h, int dtype, const void* X, const void* alpha, int64_t N, int D,
                           int B, const double* ls, const double* sf2, const double* y_mean,
                           const double* y_std, const void* Xq, int64_t M, void* mean);

/* ---- K5: posterior variance ---------------------------------------------------------------
 * var[m] = max(kss - sum_i (L^-1 k*_m)_i^2, floor) in units of the normalised targets
 * (caller multiplies by y_std^2).  kss = sf2 (+ noise for the sklearn surface).
 * work: dev scratch of at least Np * Mp elements of `dtype` (Mp = gpk_padded(M)).
 * var: dev double[Mp] (only the first M entries are meaningful).
 * Replaces: sklearn/gaussian_process/_gpr.py:454-485; gaussian_process.py:229-232.          */
GPK_API int gpk_predict_var(gpk_handle h, int dtype, const void* X, int64_t N, int D, const double* ls,
                    double sf2, const void* L, int64_t Np, int64_t ldl, const void* winv,
                    const void* Xq, int64_t M, double kss, double floor, void* work,
                    double* var);

/* ---- K5, serving form: variance through the explicit inverse factor ---------------------------------
 * gpk_trtri: W (dev Np x ldw, fp64) <- L^-1, lower triangle by tiles (recursive, fp64-MFMA GEMMs);
 * one-off N^3/3 flops after the factorisation.  work: dev double[(Np/2 + 128)^2].  Besides the lower
 * triangle it writes zeros into the 15 tiles to the right of every diagonal tile (see below).
 * gpk_tril_to_f32: fp32 copy of the lower triangle, with the same band of zeros right of the diagonal.
 * gpk_predict_var_inv: var[m] = max(kss - |W k*_m|^2, floor) in ONE GEMM launch: the tile of
 * V = W K*^T is reduced to column sums of squares in the epilogue and never written to HBM; W's
 * zero upper triangle is skipped (N^2 M flops, + 1.4 %: the tile rows of a super-tile all run to the
 * end of the longest row so that its 64 workgroups stay in lockstep and share operand panels through L2 -
 * W must therefore be ZERO above the diagonal for 16 tiles (2048 columns) from each row's diagonal tile
 * rightwards, which gpk_trtri and gpk_tril_to_f32 guarantee).  W, X, Xq of `dtype`; work: dev scratch
 * of Mp * Np elements of `dtype`; var: dev double[M] (only M entries are written).
 * Replaces the same reference lines as gpk_predict_var (sklearn/gaussian_process/_gpr.py:454-485),
 * with solve_triangular(L, K*^T) evaluated as (L^-1) K*^T.                                          */
GPK_API int gpk_trtri(gpk_handle h, const double* L, int64_t Np, int64_t ldl, const double* winv, double* W,
              int64_t ldw, double* work);
/* gpk_trtri_absmax: gpk_trtri that also leaves block_absmax[Np / 128] (dev) = max |(float)W_ij| over the lower triangle of each
 * 128-row block, accumulated as the tiles of W are written (the epilogue of the level products; no pass over W): the first
 * half of gpk_split2_rows_f64, for gpk_split2_rows_f64_absmax.  Not in batched mode.                                        */
GPK_API int gpk_trtri_absmax(gpk_handle h, const double* L, int64_t Np, int64_t ldl, const double* winv, double* W,
              int64_t ldw, double* work, float* block_absmax);
GPK_API int gpk_tril_to_f32(gpk_handle h, const double* A, int64_t Np, int64_t lda, float* Af, int64_t ldaf);
GPK_API int gpk_predict_var_inv(gpk_handle h, int dtype, const void* X, int64_t N, int D, const double* ls,
                        double sf2, const void* W, int64_t Np, int64_t ldw, const void* Xq, int64_t M,
                        double kss, double floor, void* work, double* var);

/* ---- K6a: log-marginal-likelihood terms -----------------------------------------------------
 * terms[0] = sum_{i<N} log L[i][i]; terms[1 + p] = sum_i Y[i][p] * alpha[i][p]  (host doubles).
 * Synchronises.  Replaces: sklearn/gaussian_process/_gpr.py:609-613; gaussian_process.py:250-261. */
GPK_API int gpk_lml_terms(gpk_handle h, const double* L, int64_t N, int64_t ldl, const double* Y,
                  const double* alpha, int P, double* terms);

/* ---- K6b: K^-1 and the fused LML-gradient reduction --------------------------------------------
 * gpk_potri: Kinv (dev Np x ldk) <- lower triangle of (L L^T)^-1 (trtri + W^T W on fp64 MFMA).
 * L is not modified.  work: dev double[Np * Np].
 * Replaces: cho_solve((L, True), eye(N)) at sklearn/gaussian_process/_gpr.py:627-629.           */
GPK_API int gpk_potri(gpk_handle h, const double* L, int64_t Np, int64_t ldl, const double* winv,
              double* Kinv, int64_t ldk, double* work);
/* The second half of gpk_potri when W = L^-1 is already at hand: Kinv (lower tiles) = W^T W.       */
GPK_API int gpk_wtw(gpk_handle h, const double* W, int64_t Np, int64_t ldw, double* Kinv, int64_t ldk);
/* grad[d] (d < D) = 0.5 * sum_ij Q_ij K_ij ((x_id - x_jd)/ls_d)^2, grad[D] = 0.5 * noise * tr(Q),
 * Q = alpha alpha^T - P * Kinv, K_ij = sf2 exp(-0.5 d2_ij) recomputed on the fly (the
 * N x N x D tensor sklearn builds at kernels.py:1576-1579 is never materialised).
 * grad: host double[D + 2] = [g_ls_0 .. g_ls_{D-1}, g_noise, g_sf2] with g_sf2 = 0.5 * sum_ij Q_ij K_ij
 * (the signal-variance gradient used by the package GP).  D <= 16.  Synchronises.
 * Replaces: sklearn/gaussian_process/_gpr.py:615-647 + sklearn/gaussian_process/kernels.py:1571-1580,
 * :1403-1408.                                                                                     */
GPK_API int gpk_lml_grad(gpk_handle h, const double* X, int64_t N, int D, const double* ls, double sf2,
                 double noise, const double* alpha, int P, const double* Kinv, int64_t ldk,
                 double* grad);

/* Kernel matrix of two point sets and its length-scale derivative factor, as the package GP's kernel OBJECT hands them out:
 * K[i][j] = sf2 exp(-r2_ij / 2), Q[i][j] = K[i][j] r2_ij, r2_ij = sum_d ((x1_id - x2_jd) / ls_d)^2 by exact differences
 * (the reference's norm expansion, gaussian_process.py:38, differs by ~3e-14 on the flight data).  For its isotropic kernel
 * dK/dl = Q / l and dK/dsf2 = K / sf2.  X1 dev (n1 x D), X2 dev (n2 x D), K and Q (Q may be NULL) dev (n1 x ld), ld >= n2, no
 * padding; D <= 16.  Asynchronous on the handle's stream.
 * Replaces: RBFKernel.__call__ / RBFKernel.gradient / GaussianProcess.compute_kernel_matrix,
 * quadrotor_gp_mpc/quadrotor_gp_mpc/gaussian_process.py:26-60,158-171.                                              */
GPK_API int gpk_rbf_kernel_grad(gpk_handle h, const double* X1, int64_t n1, const double* X2, int64_t n2, int D,
                        const double* ls, double sf2, double* K, double* Q, int64_t ld);

/* ---- one optimiser evaluation as one chain ------------------------------------------------------------------------
 * gpk_lml_eval: K1 (Gram of X with ls, sf2, diag_add = noise + jitter, into K), K2 (factor in place, winv), W = L^-1 (work:
 * the scratch of gpk_trtri), alpha = W^T (W Yn), the terms of gpk_lml_terms and - grad != NULL - K^-1 = W^T W (Kinv) and the
 * gradient of gpk_lml_grad: the same launches as the call-by-call route, queued back to back with ONE synchronisation
 * (call by call there are three: the pivot check, the terms, the gradient).  Device pointers except ls, terms[1 + P],
 * grad[D + 2], info (host).  Returns GPK_NOT_PD with *info as gpk_potrf (terms / grad then hold nothing).  Np = gpk_padded(N).
 * Replaces: one call of log_marginal_likelihood(theta, eval_gradient=True), sklearn/gaussian_process/_gpr.py:537-652, as the
 * optimiser of GaussianProcessRegressor.fit makes it (src/px4/simple_gp.py:167-177).                                      */
GPK_API int gpk_lml_eval(gpk_handle h, const double* X, int64_t N, int D, const double* ls, double sf2, double diag_add,
                 double noise, const double* Yn, int P, double* K, int64_t Np, double* winv, double* W, double* work,
                 double* alpha, double* Kinv, double* terms, double* grad, int* info);

/* ---- composite calls: a whole model behind the handle ------------------------------------------------------
 * For callers that are not Python: the sequencing the Python host side (device.py, gpr.py) otherwise provides,
 * as thin C++ over the building blocks above.  HOST pointers in and out; the device buffers (X, normalised
 * targets, factor, leaf inverses, inverse factor, alpha, fp32 serving copies, staging) are owned by the handle, one
 * model per handle (a new gpk_fit / gpk_import replaces it; gpk_model_release or gpk_destroy frees it).
 *
 * gpk_fit: X host (N x D), Y host (N x P) row-major fp64.  ls: n_ls = 1 (isotropic) or D (ARD) length-scales;
 *   K = sf2 exp(-d^2/2) + (noise + jitter) I (`noise`: the WhiteKernel level, `jitter`: the regressor's alpha).
 *   normalize_y != 0: targets are centred and divided by their population std per column (a zero std counts as 1).
 *   Runs K1 -> K2 -> K3 (+ the inverse factor when Np <= 32768) and evaluates the log-marginal likelihood.
 *   Returns GPK_NOT_PD as gpk_potrf does (the caller decides: sklearn raises, the package GP multiplies its noise by
 *   10, gaussian_process.py:193-201).  D <= GPK_MAX_D_PREDICT, P <= GPK_MAX_P.
 *   Replaces: GaussianProcessRegressor.fit at fixed theta, sklearn/gaussian_process/_gpr.py:271-282,343-364, as
 *   called at src/px4/simple_gp.py:170-177; GaussianProcess.fit, gaussian_process.py:173-201.
 * gpk_predict: Xq host (M x D), mean host (M x P), var host (M x P: per-output VARIANCE, already multiplied by
 *   y_std^2; NULL = means only), all of `dtype` (GPK_F64: double buffers, fp64 kernels; GPK_F32: float buffers, the
 *   fp32 serving kernels - matrix-core mean when admissible, fp16 x 2 split variance - behind the two fp32 serving
 *   gates: a model whose fp32 mean would leave 1e-4 (estimated once per model from two fp64 launches on <= 1024 training
 *   rows) is served by the fp64 kernels, and rows whose fp32 variance is below 1 % of the prior's are recomputed by the
 *   fp64 launch; GPK_F32 is a request for speed, the stated bars - mean 1e-4, std 1e-3 - hold either way).
 *   var_includes_noise != 0:
 *   k** = sf2 + noise, variance clipped at 0 (scikit-learn: Sum.diag, _gpr.py:474-485; take sqrt for its std);
 *   == 0: k** = sf2, floored at 1e-10 (gaussian_process.py:229-233).  Queries are processed in panels.
 *   Replaces: GaussianProcessRegressor.predict, _gpr.py:441-494 (src/px4/simple_gp.py:194, mpc.py:1490-1506);
 *   GaussianProcess.predict, gaussian_process.py:203-241.
 * gpk_lml: theta == NULL: *lml = the fitted model's log-marginal likelihood.  Otherwise theta = log [ls (1 or D
 *   values), noise] (n_theta = 2 or D + 1; sf2 and jitter as fitted): *lml and, if grad != NULL, its gradient with
 *   respect to theta (same layout; isotropic: summed over the features), evaluated on scratch buffers - the fitted
 *   factor is not touched.  A non-positive-definite trial matrix gives *lml = -inf, grad = 0 and GPK_OK (what an
 *   optimiser needs, _gpr.py:586-589).  Replaces: log_marginal_likelihood, _gpr.py:537-652 + kernels.py:1571-1580.
 * gpk_export / gpk_import: the model as host arrays - L (N x N row-major lower factor, zeros above the diagonal:
 *   scikit-learn's L_), alpha (N x P), y_mean / y_std (P) - e.g. to write or read the reference's model files
 *   (src/px4/train_gp_offline.py:188-194; gaussian_process.py:369-394 stores the training set and refits).  NULL
 *   outputs are skipped.  An imported model predicts; gpk_lml(theta) needs a fitted one.                        */
GPK_API int gpk_fit(gpk_handle h, const double* X, int64_t N, int D, const double* Y, int P, const double* ls, int n_ls,
            double sf2, double noise, double jitter, int normalize_y);
GPK_API int gpk_predict(gpk_handle h, const void* Xq, int64_t M, void* mean, void* var, int dtype, int var_includes_noise);
GPK_API int gpk_lml(gpk_handle h, const double* theta, int n_theta, double* lml, double* grad);
GPK_API int gpk_export(gpk_handle h, int64_t* N, int* D, int* P, double* L, double* alpha, double* y_mean, double* y_std,
               double* lml);
GPK_API int gpk_import(gpk_handle h, const double* X, int64_t N, int D, const double* L, const double* alpha, int P,
               const double* ls, int n_ls, double sf2, double noise, const double* y_mean, const double* y_std);
GPK_API int gpk_model_release(gpk_handle h);

/* ---- composite calls for B (<= GPK_MAX_BATCH) single-output models on shared inputs ----------------------------
 * The per-axis layout of src/px4/gp_trainer.py:139-179 (one scalar GP per residual component, each with its own ARD
 * length-scales and noise, all on the same X) as one object behind the handle, next to - and independent of - the
 * single model of gpk_fit.  HOST pointers in and out, fp64.
 * gpk_fit_batched: X host (N x D); Y host (N x B), column b = targets of model b; ls host (B x n_ls), n_ls = 1 or D;
 *   sf2, noise: host double[B]; jitter and normalize_y as in gpk_fit, shared.  One Gram launch per model, then ONE
 *   batched launch chain for the B factorisations, inverse factors and alpha solves (gpk_batch_begin).  info: host
 *   int[B], 0 or the 1-based index of model b's first non-positive pivot; if any is non-zero the call returns
 *   GPK_NOT_PD and the batch is not usable for prediction.
 *   Replaces: the loop over outputs of GPTrainer.train_gp_models at fixed theta, gp_trainer.py:139-179 (six times
 *   sklearn/gaussian_process/_gpr.py:271-282,343-364).
 * gpk_predict_batched: Xq host (M x D); mean host (M x B); var host (M x B: variances in target units, NULL = means
 *   only); var_includes_noise as in gpk_predict.  Up to 32 queries (N <= 16384) take gpk_predict_host_multi (one call,
 *   two launches for all models); larger batches one fused mean launch (gpk_predict_mean_multi) and one variance launch
 *   per model and panel.  Replaces: PreTrainedGP.predict_residual's loop, src/px4/pretrained_gp.py:52-98.
 * gpk_lml_batched: thetas == NULL: lml[b] = the fitted models' log-marginal likelihoods.  Otherwise thetas host
 *   (B x n_theta), row b = log [ls (1 or D values), noise] of model b: lml host double[B] and, if grad != NULL, grad
 *   host (B x n_theta), from one batched launch chain on scratch buffers (BASELINE configuration 5: the
 *   hyper-parameter step of all per-axis models at once).  A model whose trial matrix is not positive definite gets
 *   lml = -inf, grad = 0.  Replaces: B evaluations of log_marginal_likelihood, _gpr.py:537-652.
 * gpk_model_release frees this object too.                                                                         */
GPK_API int gpk_fit_batched(gpk_handle h, int B, const double* X, int64_t N, int D, const double* Y, const double* ls, int n_ls,
                    const double* sf2, const double* noise, double jitter, int normalize_y, int* info);
GPK_API int gpk_predict_batched(gpk_handle h, const double* Xq, int64_t M, double* mean, double* var, int var_includes_noise);
GPK_API int gpk_lml_batched(gpk_handle h, const double* thetas, int n_theta, double* lml, double* grad);

/* ---- building block: whole-tile GEMM on the matrix cores ---------------------------------------
 * C[m x n] = alpha * opA(A) * opB(B)^T + beta * C, m and n multiples of 128, k a multiple of 16
 * (fp64) / 32 (fp32).  ta == 0: A stored (m x k) with k contiguous; ta == 1: A stored (k x m).
 * tb == 0: B stored (n x k); tb == 1: B stored (k x n).  lower_only != 0 skips tiles strictly above
 * the diagonal.  This is the kernel behind gpk_potrf / gpk_potrs / gpk_trsm_lower_left / gpk_potri;
 * it is exported so that it can be tested and timed on its own (fp64 via v_mfma_f64_16x16x4_f64,
 * fp32 via v_mfma_f32_32x32x2_f32).  No reference counterpart other than the BLAS-3 calls inside
 * LAPACK's dpotrf/dtrsm (SciPy, sklearn/gaussian_process/_gpr.py:349,454).                          */
GPK_API int gpk_gemm_tiles(gpk_handle h, int dtype, int ta, int tb, const void* A, int64_t lda, const void* B,
                   int64_t ldb, void* C, int64_t ldc, int64_t m, int64_t n, int64_t k, double alpha,
                   double beta, int lower_only);

#ifdef __cplusplus
}
#endif
#endif /* GPK_H */
