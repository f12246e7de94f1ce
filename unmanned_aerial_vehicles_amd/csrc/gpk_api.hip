// Context management and the composite entry points of the libgpk C ABI.
#include <cstdlib>

#include "gpk_internal.h"

extern "C" const char* gpk_version(void) { return "gpk 0.1 (gfx950)"; }

extern "C" int64_t gpk_padded(int64_t n) { return (n + GPK_TILE - 1) / GPK_TILE * GPK_TILE; }

extern "C" int gpk_create(gpk_handle* out, int device) {
  if (!out) return GPK_BAD_ARG;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return GPK_HIP_ERROR;
  if (hipSetDevice(device) != hipSuccess) return GPK_HIP_ERROR;
  gpk_context* h = new gpk_context();
  h->device = device;
  if (hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess ||
      hipMalloc((void**)&h->d_info, GPK_MAX_BATCH * sizeof(int)) != hipSuccess ||

      hipMalloc((void**)&h->d_count, 2 * GPK_SMALL_MAX_MODELS * sizeof(unsigned)) != hipSuccess ||
      hipMemset(h->d_count, 0, 2 * GPK_SMALL_MAX_MODELS * sizeof(unsigned)) != hipSuccess ||
      hipHostMalloc((void**)&h->h_small, 4096, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
      hipHostGetDevicePointer((void**)&h->d_small, h->h_small, 0) != hipSuccess) {
    delete h;
    return GPK_HIP_ERROR;
  }
  h->stream = h->own_stream;
  // (the handle reads nothing from the environment: every knob is a gpk_set_option / gpk_set_option_str name)
  *out = h;
  return GPK_OK;
}

extern "C" void gpk_destroy(gpk_handle h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  gpk_model_free(h);
  if (h->scratch) (void)hipFree(h->scratch);
  if (h->d_info) (void)hipFree(h->d_info);
  if (h->d_count) (void)hipFree(h->d_count);
  if (h->d_ptile) (void)hipFree(h->d_ptile);
  if (h->d_ptile_list) (void)hipFree(h->d_ptile_list);
  if (h->h_small) (void)hipHostFree(h->h_small);
  if (h->serve_dev) (void)hipFree(h->serve_dev);
  if (h->serve_host) (void)hipHostFree(h->serve_host);
  for (auto& t : h->timed) {
    if (t.e0) (void)hipEventDestroy(t.e0);
    if (t.e1) (void)hipEventDestroy(t.e1);
  }
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  delete h;
}

extern "C" const char* gpk_last_error(gpk_handle h) { return h ? h->err.c_str() : "null handle"; }

extern "C" int gpk_set_stream(gpk_handle h, void* stream) {
  if (!h) return GPK_BAD_ARG;
  hipStream_t want = (stream == GPK_OWN_STREAM) ? h->own_stream : (hipStream_t)stream;
  // launches and allocations of the calls that follow go to the calling thread's current device: make that the
  // handle's (two handles on two GPUs in one process; the Python side calls this before every group of calls)
  GPK_CHECK_HIP(h, hipSetDevice(h->device));
  if (want == h->stream) return GPK_OK;
  // work queued on the old stream must finish before later calls may reuse the handle's scratch
  GPK_CHECK_HIP(h, hipStreamSynchronize(h->stream));
  h->stream = want;
  h->user_stream = (stream != GPK_OWN_STREAM);
  return GPK_OK;
}

extern "C" int gpk_set_option(gpk_handle h, const char* name, int value) {
  if (!h) return GPK_BAD_ARG;
  GPK_REQUIRE(h, name, "set_option: null name");
  const std::string n(name);
  if (n == "k5_split2_tile") h->k5_split2_tile = (value >= 0 && value <= 2) ? value : 0;
  else if (n == "k5_super") h->k5_super = value;
  else if (n == "small_path") h->small_path = value;
  else if (n == "trsm256") h->trsm256 = value;
  else if (n == "trtri_levels") h->trtri_levels = value;
  else if (n == "gemm_small_tiles") h->gemm_small_tiles = value;
  else if (n == "gemm_tiny_tiles") h->gemm_tiny_tiles = value;
  else if (n == "gemm_balanced") h->gemm_balanced = value;
  else if (n == "gemm_balanced_max_tiles") h->gemm_balanced_max_tiles = value;
  else if (n == "k3_stream_min_np") h->k3_stream_min_np = value;
  else if (n == "gemm_wm_f64") h->gemm_wm_f64 = value == 2 ? 2 : 4;
  else if (n == "gemm_wm_f32") h->gemm_wm_f32 = value == 2 ? 2 : 4;
  else if (n == "gemm_log") h->gemm_log = value;
  else if (n == "debug_fill") h->debug_fill = value ? 1 : 0;
  else if (n == "ptile_slots") h->ptile_slots_override = value > 0 ? value : 0;
  else if (n == "ptile") h->ptile = value;
  else if (n == "ptile_max_np") h->ptile_max_np = value;
  else if (n == "ptile_prog_max_nt") h->ptile_prog_max_nt = value;
  else if (n == "ptile_inv_max_np") h->ptile_inv_max_np = value;
  else if (n == "ptile_single_max_nt") h->ptile_single_max_nt = value;
  else if (n == "ptile_xcd") h->ptile_xcd = value;
  else if (n == "ptile_grp_rows") h->ptile_grp_rows = value;
  else if (n == "ptile_grp_cols") h->ptile_grp_cols = value;
  else if (n == "ptile_xcd_min_nt") h->ptile_xcd_min_nt = value;
  else if (n == "ptile_sr") h->ptile_sr = value;
  else if (n == "ptile_sr_max_nt") h->ptile_sr_max_nt = value;
  else if (n == "ptile_prog_rows") h->ptile_prog_rows = value >= 8 ? 8 : value >= 1 ? value : 1;
  else { h->err = "bad argument: unknown option " + n; return GPK_BAD_ARG; }
  return GPK_OK;
}

extern "C" int gpk_set_option_str(gpk_handle h, const char* name, const char* value) {
  if (!h) return GPK_BAD_ARG;
  GPK_REQUIRE(h, name, "set_option_str: null name");
  const std::string n(name);
  if (n == "ptile_trace_path") h->ptile_trace_request = value ? value : "";
  else { h->err = "bad argument: unknown option " + n; return GPK_BAD_ARG; }
  return GPK_OK;
}

extern "C" int gpk_timing(gpk_handle h, int enable) {
  if (!h) return GPK_BAD_ARG;
  if (enable && h->timed.empty()) {
    GPK_CHECK_HIP(h, hipSetDevice(h->device));
    h->timed.resize(GPK_TIMING_RING);
    for (auto& t : h->timed) {
      GPK_CHECK_HIP(h, hipEventCreate(&t.e0));
      GPK_CHECK_HIP(h, hipEventCreate(&t.e1));
    }
  }
  h->timing = enable ? 1 : 0;
  h->timed_count = 0;
  return GPK_OK;
}

extern "C" int gpk_kernel_times(gpk_handle h, int tag, double* ms, int max_n, int* n_out) {
  if (!h) return GPK_BAD_ARG;
  GPK_REQUIRE(h, ms && n_out && max_n >= 1, "kernel_times: null pointer");
  *n_out = 0;
  if (h->timed.empty()) return GPK_OK;
  GPK_CHECK_HIP(h, hipStreamSynchronize(h->stream));
  const long long have = h->timed_count < GPK_TIMING_RING ? h->timed_count : GPK_TIMING_RING;
  // oldest first among the launches still in the ring
  for (long long i = h->timed_count - have; i < h->timed_count && *n_out < max_n; ++i) {
    const auto& t = h->timed[(size_t)(i % GPK_TIMING_RING)];
    if (t.tag != tag) continue;
    float f = 0.f;
    GPK_CHECK_HIP(h, hipEventElapsedTime(&f, t.e0, t.e1));
    ms[(*n_out)++] = (double)f;
  }
  return GPK_OK;
}

extern "C" int gpk_batch_begin(gpk_handle h, int count) {
  if (!h) return GPK_BAD_ARG;
  GPK_REQUIRE(h, count >= 1 && count <= GPK_MAX_BATCH, "batch_begin: count must be in [1, 8]");
  GPK_REQUIRE(h, h->batch == 1 && h->bbufs.empty(), "batch_begin: a batch is already open");
  h->batch = count;
  return GPK_OK;
}

extern "C" int gpk_batch_buffer(gpk_handle h, const void* base, int64_t stride_bytes) {
  if (!h) return GPK_BAD_ARG;
  GPK_REQUIRE(h, base && stride_bytes > 0 && stride_bytes % 16 == 0, "batch_buffer: stride must be a positive multiple of 16");
  h->bbufs.push_back({(const char*)base, (long long)stride_bytes});
  return GPK_OK;
}

extern "C" int gpk_batch_end(gpk_handle h) {
  if (!h) return GPK_BAD_ARG;
  h->batch = 1;
  h->bbufs.clear();
  return GPK_OK;
}

extern "C" int gpk_synchronize(gpk_handle h) {
  if (!h) return GPK_BAD_ARG;
  GPK_CHECK_HIP(h, hipStreamSynchronize(h->stream));
  return GPK_OK;
}

int gpk_scratch(gpk_handle h, size_t bytes, void** out) {
  if (bytes > h->scratch_bytes) {
    GPK_CHECK_HIP(h, hipSetDevice(h->device));
    // growing: wait for users of the old block, then replace it
    GPK_CHECK_HIP(h, hipStreamSynchronize(h->stream));
    if (h->scratch) GPK_CHECK_HIP(h, hipFree(h->scratch));
    h->scratch = nullptr;
    h->scratch_bytes = 0;
    const size_t want = (bytes + (1u << 20) - 1) & ~(size_t)((1u << 20) - 1);
    GPK_CHECK_HIP(h, hipMalloc(&h->scratch, want));
    h->scratch_bytes = want;
  }
  // debugging aid (the tests run with it): whoever asked for scratch must write it before reading it
  if (h->debug_fill && bytes > 0) GPK_CHECK_HIP(h, hipMemsetAsync(h->scratch, 0xFF, bytes, h->stream));
  *out = h->scratch;
  return GPK_OK;
}

extern "C" int gpk_predict_var(gpk_handle h, int dtype, const void* X, int64_t N, int D, const double* ls,
                               double sf2, const void* L, int64_t Np, int64_t ldl, const void* winv, const void* Xq,
                               int64_t M, double kss, double floor_, void* work, double* var) {
  if (!h) return GPK_BAD_ARG;
  GPK_REQUIRE(h, X && L && winv && Xq && work && var, "predict_var: null pointer");
  GPK_REQUIRE(h, N >= 1 && M >= 1 && Np == gpk_padded(N), "predict_var: Np must equal gpk_padded(N)");
  const int64_t Mp = gpk_padded(M);
  // B = K*^T (Np x Mp), V = L^-1 B in place, var = kss - colsumsq(V)
  GPK_TRY(gpk_cross_gram_t(h, dtype, X, N, Xq, M, D, ls, sf2, work, Mp));
  GPK_TRY(gpk_trsm_lower_left(h, dtype, L, Np, ldl, winv, work, Mp, Mp));
  // column sums of squares land in var[0..Mp) and are finalised in place
  GPK_TRY(gpk_colsumsq(h, dtype, work, Np, Mp, Mp, var));
  return gpk_var_finalize(h, var, M, kss, floor_, var);
}

extern "C" int gpk_predict_var_inv(gpk_handle h, int dtype, const void* X, int64_t N, int D, const double* ls,
                                   double sf2, const void* W, int64_t Np, int64_t ldw, const void* Xq, int64_t M,
                                   double kss, double floor_, void* work, double* var) {
  if (!h) return GPK_BAD_ARG;
  GPK_REQUIRE(h, X && W && Xq && work && var, "predict_var_inv: null pointer");
  GPK_REQUIRE(h, N >= 1 && M >= 1 && Np == gpk_padded(N) && ldw >= Np, "predict_var_inv: Np must equal gpk_padded(N)");
  GPK_REQUIRE(h, dtype == GPK_F32 || dtype == GPK_F64, "predict_var_inv: bad dtype");
  const int64_t Mp = gpk_padded(M);
  // Kq (Mp x Np, query-major, k contiguous) = k(Xq, X); zero in the padding
  GPK_TRY(gpk_cross_gram_t(h, dtype, Xq, M, X, N, D, ls, sf2, work, Np));
  // one launch: V = W Kq^T tile by tile (W lower: k < row-tile end), V never stored, only the per-tile
  // column sums of squares: partial[tile_row][m]
  GemmArgs g = gemm_args(W, ldw, 0, work, Np, 0, nullptr, Mp, (int)Np, (int)Mp, (int)Np, 1.0, 0.0);
  g.ke0 = GPK_TILE;
  g.ke_row = GPK_TILE;
  g.epilogue = 1;
  g.k_super = h->k5_super;   // W is zero right of the diagonal for GPK_ZERO_BAND_TILES - 1 tiles (gpk_trtri, gpk_tril_to_f32)
  g.heavy_first = 1;   // row tile tm costs (tm + 1) k-blocks: start the long ones first
  const int ntm = (int)(Np / gpk_gemm_tile(h, g));       // one partial row per tile row of the launch
  void* partial = nullptr;
  GPK_TRY(gpk_scratch(h, (size_t)ntm * Mp * sizeof(double), &partial));
  g.C = partial;
  gpk_time_begin(h, GPK_TIMED_K5);
  const int rc_gemm = gpk_gemm(h, dtype, g);
  gpk_time_end(h);
  GPK_TRY(rc_gemm);
  return gpk_colsum_finalize(h, (const double*)partial, ntm, Mp, M, kss, floor_, var);
}

// ---- one-call serving for the control loop: host queries in, host mean / variance out ------------------
// Grow the serving staging blocks (pinned, device-mapped host block; device work block) to at least these sizes.
static int serve_reserve(gpk_handle h, size_t host_need, size_t dev_need) {
  if (host_need <= h->serve_host_bytes && dev_need <= h->serve_dev_bytes) return GPK_OK;
  GPK_CHECK_HIP(h, hipSetDevice(h->device));
  GPK_CHECK_HIP(h, hipStreamSynchronize(h->stream));
  if (host_need > h->serve_host_bytes) {
    if (h->serve_host) GPK_CHECK_HIP(h, hipHostFree(h->serve_host));
    h->serve_host = nullptr; h->serve_host_bytes = 0;
    const size_t want = (host_need + 65535) & ~(size_t)65535;
    GPK_CHECK_HIP(h, hipHostMalloc(&h->serve_host, want, hipHostMallocMapped | hipHostMallocCoherent));
    h->serve_host_bytes = want;
  }
  if (dev_need > h->serve_dev_bytes) {
    if (h->serve_dev) GPK_CHECK_HIP(h, hipFree(h->serve_dev));
    h->serve_dev = nullptr; h->serve_dev_bytes = 0;
    const size_t want = (dev_need + (1u << 20) - 1) & ~(size_t)((1u << 20) - 1);
    GPK_CHECK_HIP(h, hipMalloc(&h->serve_dev, want));
    h->serve_dev_bytes = want;
  }
  return GPK_OK;
}

extern "C" int gpk_predict_host(gpk_handle h, const double* X, const double* alpha, int64_t N, int D, int P,
                                const double* ls, double sf2, const double* y_mean, const double* y_std,
                                const double* W, int64_t Np, int64_t ldw, double kss, double floor_,
                                const double* Xq_host, int64_t M, double* mean_host, double* var_host) {
  if (!h) return GPK_BAD_ARG;
  GPK_REQUIRE(h, X && alpha && ls && y_mean && y_std && Xq_host && mean_host, "predict_host: null pointer");
  GPK_REQUIRE(h, N >= 1 && M >= 1 && M <= GPK_HOST_MAX_M, "predict_host: M must be in [1, GPK_HOST_MAX_M]");
  GPK_REQUIRE(h, !var_host || (W && Np == gpk_padded(N) && ldw >= Np), "predict_host: variance needs the inverse factor");
  GPK_REQUIRE(h, h->batch == 1, "predict_host: not available in batched mode");
  const int64_t Mp = gpk_padded(M);
  // up to 32 queries: the two small-batch launches; 33..64: the same twice (4 launches instead of the general chain's 7)
  const bool small = h->small_path && M <= 2 * GPK_SMALL_MAX_M && gpk_small_ok(gpk_padded(N), D, P, M < GPK_SMALL_MAX_M ? M : GPK_SMALL_MAX_M);
  // pinned host block [Xq | pad][mean | var | pad]; device block [Xq | pad][work: the K* panel of the variance GEMM,
  // or the small-batch kernels' K* and shares]
  const size_t nq = ((size_t)M * D + 15) & ~(size_t)15, nm = (size_t)M * P, nv = (size_t)M;
  const size_t nout_pad = (nm + nv + 15) & ~(size_t)15;
  const size_t host_need = (nq + nout_pad) * sizeof(double);
  const size_t work_need = small ? gpk_small_work_doubles(gpk_padded(N), 1) : (var_host ? (size_t)Mp * Np : 0);
  const size_t dev_need = (nq + work_need) * sizeof(double);
  GPK_TRY(serve_reserve(h, host_need, dev_need));
  // The staging block is pinned, coherent host memory mapped into the device's address space: mean / variance are
  // written by the kernels straight into it (a few hundred bytes over PCIe): no download command, one stream
  // synchronisation.  Small batches (<= 32 queries): the kernels read the queries from it as well -- two launches,
  // no copy command (gpk_small.hip).  Otherwise the queries go to HBM with one async copy (every workgroup re-reads
  // them) ahead of the general chain.
  double* hq = (double*)h->serve_host;
  double* hout = hq + nq;                      // [mean | var]
  double* dq = (double*)h->serve_dev;
  double* dwork = dq + nq;
  if (h->debug_fill && work_need > 0) GPK_CHECK_HIP(h, hipMemsetAsync(dwork, 0xFF, work_need * sizeof(double), h->stream));
  memcpy(hq, Xq_host, (size_t)M * D * sizeof(double));
  if (small) {
    for (int64_t m0 = 0; m0 < M; m0 += GPK_SMALL_MAX_M) {
      const int64_t mc = M - m0 < GPK_SMALL_MAX_M ? M - m0 : GPK_SMALL_MAX_M;
      GPK_TRY(gpk_small_predict(h, 1, &X, &alpha, N, D, P, ls, &sf2, y_mean, y_std, &W, gpk_padded(N), ldw, &kss, floor_,
                                hq + m0 * D, mc, dwork, hout + m0 * P, var_host ? hout + nm + m0 : nullptr));
    }
  } else {
    GPK_CHECK_HIP(h, hipMemcpyAsync(dq, hq, (size_t)M * D * sizeof(double), hipMemcpyHostToDevice, h->stream));
    GPK_TRY(gpk_predict_mean(h, GPK_F64, X, alpha, N, D, P, ls, sf2, y_mean, y_std, dq, M, hout));
    if (var_host) GPK_TRY(gpk_predict_var_inv(h, GPK_F64, X, N, D, ls, sf2, W, Np, ldw, dq, M, kss, floor_, dwork, hout + nm));
  }
  GPK_CHECK_HIP(h, hipStreamSynchronize(h->stream));
  memcpy(mean_host, hout, nm * sizeof(double));
  if (var_host) memcpy(var_host, hout + nm, nv * sizeof(double));
  return GPK_OK;
}

extern "C" int gpk_predict_host_multi(gpk_handle h, int B, const double* const* X, const double* const* alpha, int64_t N,
                                      int D, const double* ls, const double* sf2, const double* y_mean, const double* y_std,
                                      const double* const* W, int64_t Np, int64_t ldw, const double* kss, double floor_,
                                      const double* Xq_host, int64_t M, double* mean_host, double* var_host) {
  if (!h) return GPK_BAD_ARG;
  GPK_REQUIRE(h, X && alpha && ls && sf2 && y_mean && y_std && Xq_host && mean_host, "predict_host_multi: null pointer");
  GPK_REQUIRE(h, B >= 1 && B <= GPK_SMALL_MAX_MODELS, "predict_host_multi: 1..8 models");
  GPK_REQUIRE(h, N >= 1 && gpk_small_ok(gpk_padded(N), D, 1, M), "predict_host_multi: needs M <= 32, D <= 16, N <= 16384");
  GPK_REQUIRE(h, !var_host || (W && kss && Np == gpk_padded(N) && ldw >= Np), "predict_host_multi: variance needs the inverse factors");
  GPK_REQUIRE(h, h->batch == 1, "predict_host_multi: not available in batched mode");
  const size_t nq = ((size_t)M * D + 15) & ~(size_t)15, nm = (size_t)B * M, nv = (size_t)B * M;
  const size_t nout_pad = (nm + nv + 15) & ~(size_t)15;
  GPK_TRY(serve_reserve(h, (nq + nout_pad) * sizeof(double), (nq + gpk_small_work_doubles(gpk_padded(N), B)) * sizeof(double)));
  double* hq = (double*)h->serve_host;
  double* hout = hq + nq;                      // [mean (B x M) | var (B x M)]
  double* dwork = (double*)h->serve_dev + nq;
  memcpy(hq, Xq_host, (size_t)M * D * sizeof(double));
  GPK_TRY(gpk_small_predict(h, B, X, alpha, N, D, 1, ls, sf2, y_mean, y_std, W, gpk_padded(N), ldw, kss, floor_, hq, M, dwork,
                            hout, var_host ? hout + nm : nullptr));
  GPK_CHECK_HIP(h, hipStreamSynchronize(h->stream));
  memcpy(mean_host, hout, nm * sizeof(double));
  if (var_host) memcpy(var_host, hout + nm, nv * sizeof(double));
  return GPK_OK;
}

extern "C" int gpk_gemm_tiles(gpk_handle h, int dtype, int ta, int tb, const void* A, int64_t lda, const void* B,
                              int64_t ldb, void* C, int64_t ldc, int64_t m, int64_t n, int64_t k, double alpha,
                              double beta, int lower_only) {
  if (!h) return GPK_BAD_ARG;
  GPK_REQUIRE(h, A && B && C, "gemm_tiles: null pointer");
  GPK_REQUIRE(h, dtype == GPK_F32 || dtype == GPK_F64, "gemm_tiles: bad dtype");
  GPK_REQUIRE(h, m < (1ll << 31) && n < (1ll << 31) && k < (1ll << 31), "gemm_tiles: size too large");
  GemmArgs g = gemm_args(A, lda, ta ? 1 : 0, B, ldb, tb ? 1 : 0, C, ldc, (int)m, (int)n, (int)k, alpha, beta);
  g.lower_only = lower_only ? 1 : 0;
  return gpk_gemm(h, dtype, g);
}
