// Internal declarations shared by the libgpk translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/gpk.h"

struct gpk_model;    // the composite calls' model (gpk_model.hip)
struct gpk_bmodel;   // B single-output models on shared inputs (gpk_fit_batched, gpk_model.hip)

struct gpk_context {
  int device = 0;
  gpk_model* model = nullptr;        // owned: gpk_fit / gpk_import create it, gpk_destroy / gpk_model_release free it
  gpk_bmodel* bmodel = nullptr;      // owned: gpk_fit_batched creates it, gpk_destroy / gpk_model_release free it
  hipStream_t stream = nullptr;      // stream kernels are launched on
  hipStream_t own_stream = nullptr;  // created by gpk_create
  bool user_stream = false;
  std::string err;
  // growable device scratch owned by the handle (small / medium temporaries)
  void* scratch = nullptr;
  size_t scratch_bytes = 0;
  int* d_info = nullptr;        // device int for potrf pivot failures
  double* h_small = nullptr;    // 4 KiB of pinned, device-mapped host memory: the reductions' results (LML terms, gradient sums, status words)
  double* d_small = nullptr;    // ... its device address: the kernels write there, the host reads h_small after the synchronisation - no copy command
  unsigned* d_count = nullptr;  // 2 x 8 zero-initialised ticket counters (last-workgroup reductions, gpk_small.hip)
  // staging of gpk_predict_host: device block [Xq | mean | var | K* work] and its pinned host mirror [Xq | mean | var]
  void* serve_dev = nullptr;
  size_t serve_dev_bytes = 0;
  void* serve_host = nullptr;
  size_t serve_host_bytes = 0;
  int gemm_wm_f64 = 4;          // wave rows per GEMM workgroup (2 or 4); 4 = 512 threads, 4 waves/SIMD
  int gemm_wm_f32 = 4;
  int gemm_small_tiles = 1024;   // launches with fewer 128 x 128 tiles than this run on 64 x 64 tiles
  int gemm_tiny_tiles = 320;     // fp64 launches with fewer 128 x 128 tiles than this run on 32 x 32 tiles (option gemm_tiny_tiles; 0: never);
                                 // measured (profiles/r05_gemm_tiny_ab.log): W^T W at N = 1024 59 -> 33 us, 2048 108 -> 84 us, trtri 1024 117 -> 74 us;
                                 // with 1024 the products of N >= 4096 lose
  int gemm_balanced = 1;         // launches whose tiles differ in k-range: balanced persistent tile schedule
  long long gemm_balanced_max_tiles = 32768;   // ... up to this many tiles per launch (all problems of a batch)
  int cus = 0;                   // compute units of the device (read once)
  int trtri_levels = 1;      // gpk_trtri: one batched launch per level for power-of-two tile counts
  int trsm256 = 1;           // potrf: fused 256-wide base of the triangular solve
  int small_path = 1;        // gpk_predict_host: two-launch small-batch kernels (option small_path = 0 disables)
  int k5_super = 1;          // K5: lockstep super-tiles (option k5_super = 0 disables)
  int k3_stream_min_np = 512;    // gpk_potrs_inv: streaming matrix-vector passes from this padded size up (P <= 6); measured faster than the two
                                 // 128-column tile GEMMs from there on (N = 4096: 0.11 against 0.35 ms, profiles/r04_k3_ab.log)
  int k5_split2_tile = 0;    // fp16 x 2 variance launch: 0 = the tallest tile (512 / 256 / 128 x 128) that still comes in >= 512
                             // tiles; 1 = always 128 x 128; 2 = 512 x 128 whenever Np % 512 == 0
  int ptile = 1;             // gpk_potrf: one persistent launch (gpk_ptile.hip) for 512 <= Np <= ptile_max_np
  int ptile_max_np = 24576;  // (above, the recursion: halves that are one launch each + GEMMs.  One launch against that at 20 480 rows 46.7 / 48.6 ms,
                             // 24 576 78.6 / 79.2, 32 768 179.4 / 172.5, 40 960 347.4 / 340.6: profiles/r05_ptile_large.log; 16 384 until round 5)
  int ptile_prog_max_nt = 128;  // ... up to this many tile columns the two tiles under a diagonal tile follow that tile's factorisation 16
                             // columns at a time instead of waiting for the whole inverse (0: never)
  int ptile_inv_max_np = 4608;  // gpk_lml_eval: up to this padded size the inverse factor's tiles are tasks of the same launch (0: never)
  int ptile_prog_rows = 8;   // ... how many tiles under the diagonal one do so (1 .. 8).  Measured (profiles/r05_ptile_followers_ab.log): N = 4096
                             // 1.69 ms with none, 1.37 with one, 1.22 with two, 1.18 with four, 1.15 with eight, the same with twelve / sixteen
  int* d_ptile = nullptr;    // its ticket counter, abort word and per-tile-row progress counters
  int ptile_slots = 512;     // workgroups that fit the device at two per CU
  int ptile_sr = 1, ptile_sr_max_nt = 36;   // ... and up to this many tile columns the 256-register build (two k-tiles in flight in the off-diagonal
                             // k-loops: gpk_ptile.hip, SR); ptile_sr = 0: the 128-register build for everything
  int ptile_single_max_nt = 96;   // ... up to this many tile columns the launch keeps ONE workgroup per CU
  int ptile_xcd = 0;         // 1: one task queue per XCD, tile rows dealt round-robin; 2: groups of rows x columns tiles per queue; 0: ONE
                             // global ticket counter - the default: measured, neither dealing raises the L2 hit rate (the tasks of an XCD
                             // do not walk k in step) and both are 0-4 % slower (profiles/r05_ptile_xcd_ab.log)
  int ptile_xcd_min_nt = 56; // ... from this many tile columns up (below, the launch is bound by the diagonal chain, not by L2 traffic)
  int ptile_grp_rows = 8, ptile_grp_cols = 4;   // ptile_xcd = 2: groups of rows x columns tiles per queue entry block
  int* d_ptile_list = nullptr;         // the queues' task lists (device), kept for the last shape
  size_t ptile_list_cap = 0;
  std::vector<int> ptile_list_host;
  long long ptile_list_key = -1;
  int ptile_launches = 0;    // one-launch factorisations issued by the current gpk_potrf (their abort words are checked at its end)
  std::string ptile_trace_request;   // option "ptile_trace_path" (debugging aid): the NEXT one-launch factorisation writes its per-task time stamps there
  int ptile_slots_override = 0;      // option "ptile_slots" (experiments): resident workgroups of the launch, 0 = the rule below
  std::string ptile_trace_path;   // ... while that launch is in flight
  long long ptile_trace_n = 0;
  int debug_fill = 0;        // option debug_fill: the handle's scratch is overwritten with 0xFF bytes (NaN) at every request
  int gemm_log = 0;          // option gemm_log = 1: log every tile-GEMM launch to stderr (profiling aid)
  // gpk_timing: HIP-event brackets around the dominant launches (K5 variance GEMM, K1 Gram kernel), a ring of pairs
  struct TimedLaunch { hipEvent_t e0 = nullptr, e1 = nullptr; int tag = 0; };
  int timing = 0;
  std::vector<TimedLaunch> timed;   // GPK_TIMING_RING entries once enabled
  long long timed_count = 0;        // launches bracketed so far (ring position = count % size)
  // batched mode (gpk_batch_begin .. gpk_batch_end): `batch` same-shaped problems per call.  Pointers passed
  // to the entry points address problem 0; a pointer that falls inside a registered buffer advances by that
  // buffer's stride per problem, any other pointer is shared by all problems.
  struct BatchBuf { const char* base; long long stride; };
  int batch = 1;
  std::vector<BatchBuf> bbufs;
};

#define GPK_CHECK_HIP(h, call)                                                          \
  do {                                                                                  \
    hipError_t e_ = (call);                                                             \
    if (e_ != hipSuccess) {                                                             \
      (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                     \
      return GPK_HIP_ERROR;                                                             \
    }                                                                                   \
  } while (0)

#define GPK_REQUIRE(h, cond, msg)                                                       \
  do {                                                                                  \
    if (!(cond)) {                                                                      \
      (h)->err = std::string("bad argument: ") + (msg);                                 \
      return GPK_BAD_ARG;                                                               \
    }                                                                                   \
  } while (0)

#define GPK_LAUNCH_CHECK(h)                                                             \
  do {                                                                                  \
    hipError_t e_ = hipGetLastError();                                                  \
    if (e_ != hipSuccess) {                                                             \
      (h)->err = std::string("kernel launch: ") + hipGetErrorString(e_);                \
      return GPK_HIP_ERROR;                                                             \
    }                                                                                   \
  } while (0)

#define GPK_TRY(expr)                                                                   \
  do {                                                                                  \
    int rc_ = (expr);                                                                   \
    if (rc_ != GPK_OK) return rc_;                                                      \
  } while (0)

int gpk_scratch(gpk_handle h, size_t bytes, void** out);

// ---- one-launch tile Cholesky (gpk_ptile.hip) -----------------------------------------
constexpr size_t GPK_PTILE_CTRL_INTS = 16 + 8 * 512 + 1024 + 10 * 8 * 512;   // ticket, abort, padding; GPK_MAX_BATCH x (Np / 128 <= 512) row
                                                        // counters; one pause word per CU; per tile row: 16-column steps of the diagonal
                                                        // tile / of the tile left of it / of the tile left of that one published so far
// *used = 1: the launch was issued (the caller synchronises, reads info and calls gpk_potrf_ptile_check); 0: not served
// wband / ldw (optional): the band of zeros right of the diagonal tiles of this matrix is written by the launch's own preparation
// (gpk_trtri's zero_band_kernel, for the caller that goes on to the inverse factor); zero_info: so are the pivot words
int gpk_potrf_ptile(gpk_handle h, double* A, int64_t Np, int64_t lda, double* winv, int row0, int* used, double* wt = nullptr,
                    double* wband = nullptr, int64_t ldw = 0, int zero_info = 0);
int gpk_potrf_ptile_check(gpk_handle h, int gave_up = -1);
void gpk_model_free(gpk_handle h);   // gpk_model.hip
// the launches of gpk_potrf / gpk_lml_terms / gpk_lml_grad without their synchronisations (gpk_lml_eval)
int gpk_potrf_enqueue(gpk_handle h, double* A, int64_t Np, int64_t lda, double* winv);
int gpk_potrf_finish(gpk_handle h, const int* hinfo_all, int* info, int gave_up = -1);
// status words of an evaluation chain: ints [0, 8] of h->d_small + GPK_STATUS_OFF receive the pivot failures (h->batch ints) and the
// one-launch factorisation's "gave up" flag (written by the terms launch: gpk_lml_terms_enqueue(..., with_status = 1))
constexpr int GPK_STATUS_OFF = 480;
// max |(float)W_ij| over the lower triangle per 128-row block, as float bits (first pass of gpk_split2_rows_f64)
int gpk_tril_block_absmax_f64_enqueue(gpk_handle h, const double* W, int64_t n, int64_t ld, unsigned* out);
int gpk_potrf_trtri_enqueue(gpk_handle h, double* A, int64_t Np, int64_t lda, double* winv, double* W, int64_t ldw, double* wt,
                            int* used);   // factor + inverse factor as one persistent launch (small matrices), or *used = 0
int gpk_lml_terms_enqueue(gpk_handle h, const double* L, int64_t N, int64_t ldl, const double* Y, const double* alpha, int P,
                          double* dout, int with_status = 0);   // with_status: + the chain's status words, in the same launch
int gpk_lml_grad_enqueue(gpk_handle h, const double* X, int64_t N, int D, const double* ls, double sf2, const double* alpha,
                         int P, const double* Kinv, int64_t ldk, double* dout);
// B single-output evaluations on shared X as ONE chain with ONE synchronisation (gpk_lml_batched): the buffers hold the B
// problems one behind the other (K, W, Kinv: Np * Np doubles apart; winv: Np * 128; T: tsz; Yn, alpha: Ne).  K holds the
// Gram matrices on entry.  terms[2 b .. 2 b + 1], grads[b * (D + 2) ..] (gpk_lml_grad's order), info[b] (pivot failures).
int gpk_lml_chain_batched(gpk_handle h, int B, const double* X, int64_t N, int D, const double* ls /* B x D */, const double* sf2,
                          const double* noise, const double* Yn, int64_t Ne, double* K, int64_t Np, double* winv, double* W,
                          double* T, size_t tsz, double* alpha, double* Kinv, double* terms, double* grads, int* info);

// Event brackets of gpk_timing (no-ops unless enabled): record the first event, launch, record the second.
constexpr int GPK_TIMING_RING = 64;
inline void gpk_time_begin(gpk_handle h, int tag) {
  if (!h->timing) return;
  auto& t = h->timed[(size_t)(h->timed_count % GPK_TIMING_RING)];
  t.tag = tag;
  (void)hipEventRecord(t.e0, h->stream);
}
inline void gpk_time_end(gpk_handle h) {
  if (!h->timing) return;
  auto& t = h->timed[(size_t)(h->timed_count % GPK_TIMING_RING)];
  (void)hipEventRecord(t.e1, h->stream);
  ++h->timed_count;
}

// byte stride between consecutive problems of a batch for the buffer `p` points into (0: shared / not batched)
inline long long gpk_bstride(gpk_handle h, const void* p) {
  if (h->batch <= 1) return 0;
  const char* c = (const char*)p;
  for (const auto& b : h->bbufs)
    if (c >= b.base && c < b.base + b.stride) return b.stride;
  return 0;
}

// ---- dense GEMM on MFMA (gpk_gemm.hip) ---------------------------------------------
// C[m x n] = alpha * opA(A) * opB(B)^T + beta * C on whole 128x128 tiles.
//   ta == 0: A stored (m x k), k contiguous;  ta == 1: A stored (k x m), m contiguous.
//   tb == 0: B stored (n x k), k contiguous;  tb == 1: B stored (k x n), n contiguous.
// m, n multiples of 128; k multiple of 16.  lower_only skips tiles strictly above the
// diagonal.  Per-tile k range: [kb0 + kb_row * tile_row + kb_col * tile_col,
// ke0 + ke_row * tile_row + ke_col * tile_col) clipped to [0, k) (ke0 < 0 means "k").
// tiles from the diagonal tile rightwards that gpk_trtri / gpk_tril_to_f32 zero; a k_super launch may read that far
// beyond a row's own k-range (a super-tile spans at most two bands of 8 tile rows)
constexpr int GPK_ZERO_BAND_TILES = 16;

struct GemmArgs {
  const void* A;
  const void* B;
  void* C;
  int64_t lda, ldb, ldc;
  int m, n, k;
  int ta, tb;
  double alpha, beta;
  int lower_only;
  int kb0, kb_row, kb_col, ke0, ke_row, ke_col;
  int heavy_first;  // process tile rows in reverse order (use when the k-range grows with the tile row)
  int k_super;      // every row of an 8-row super-tile takes the k-range of its longest row (the operand must be
                    // zero beyond each row's own range): the 64 workgroups of a super-tile then run in lockstep
  int epilogue;   // 0: store C;  1: C (fp64, ld = ldc) [tile_row][col] = sum over the tile's rows of (alpha*acc)^2
                  // 2: store C (beta = 0) and atomicMax |(float)C_ij| into amax[128-row block of the matrix at amax_base, ld = ldc]
  unsigned* amax;
  const void* amax_base;
  // nbatch > 0: that many independent products in one launch (second grid dimension), operand i at base + i * stride
  // (bytes); in the handle's batched mode every problem of the batch runs all of them.
  int nbatch;
  long long sA, sB, sC;
};
inline GemmArgs gemm_args(const void* A, int64_t lda, int ta, const void* B, int64_t ldb, int tb,
                          void* C, int64_t ldc, int m, int n, int k, double alpha, double beta) {
  GemmArgs g{};
  g.A = A; g.B = B; g.C = C; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.m = m; g.n = n; g.k = k; g.ta = ta; g.tb = tb; g.alpha = alpha; g.beta = beta;
  g.lower_only = 0; g.kb0 = 0; g.kb_row = 0; g.kb_col = 0; g.ke0 = -1; g.ke_row = 0; g.ke_col = 0; g.epilogue = 0; g.heavy_first = 0; g.k_super = 0;
  g.nbatch = 0; g.sA = 0; g.sB = 0; g.sC = 0; g.amax = nullptr; g.amax_base = nullptr;
  return g;
}
int gpk_gemm(gpk_handle h, int dtype, const GemmArgs& g);
int gpk_gemm_tile(gpk_handle h, const GemmArgs& g);   // 128 or 64: the tile edge gpk_gemm will pick

// ---- small-batch serving kernels (gpk_small.hip) -------------------------------------
constexpr int GPK_SMALL_MAX_M = 32;          // queries per call
constexpr int64_t GPK_SMALL_MAX_NP = 16384;  // padded training rows
constexpr int GPK_SMALL_MAX_MODELS = 8;      // single-output models served by one call
bool gpk_small_ok(int64_t Np, int D, int P, int64_t M);
size_t gpk_small_work_doubles(int64_t Np, int B);   // device work area: per model K* (32 x Np) + the workgroups' shares
// B models x P outputs each (B > 1: P == 1): mean (B, M, P) and, if var_out, variance (B, M) of M <= 32 fp64 queries
// shared by the models; Xq / mean_out / var_out may be mapped host memory.  X / alpha / W: B device pointers;
// ls: B x D; sf2, kss: B; y_mean, y_std: B * P.
int gpk_small_predict(gpk_handle h, int B, const double* const* X, const double* const* alpha, int64_t N, int D, int P,
                      const double* ls, const double* sf2, const double* y_mean, const double* y_std,
                      const double* const* W, int64_t Np, int64_t ldw, const double* kss, double floor_,
                      const double* Xq, int64_t M, double* work, double* mean_out, double* var_out);

// K* straight into the fragment-order fp16 x 2 split layout (gpk_gram.hip); D <= 16
int gpk_cross_split2(gpk_handle h, const float* Xq, int64_t M, const float* X, int64_t N, int D, const double* ls,
                     double sf2, double scale, void* dst);
int gpk_var_finalize(gpk_handle h, const double* ss, int64_t M, double kss, double floor_, double* var);
int gpk_colsum_reduce(gpk_handle h, const double* partial, int S, int64_t Mp, double* out);
int gpk_colsum_finalize(gpk_handle h, const double* partial, int S, int64_t Mp, int64_t M, double kss, double floor_,
                        double* var);   // both of the above in one launch (entries M..Mp of var are left alone)
int gpk_colsum_finalize_packed(gpk_handle h, const double* partial, int S, int64_t Mp, int64_t M, double kss, double floor_,
                               const float* mean, int P, const double* y_std, double recheck_below, unsigned* low_count,
                               double* out);   // + [mean | var y_std^2] rows
