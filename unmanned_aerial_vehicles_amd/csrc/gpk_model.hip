// Composite entry points of the libgpk C ABI: a whole GP model behind the handle - gpk_fit, gpk_predict, gpk_lml,
// gpk_export, gpk_import, gpk_model_release - for callers that are not Python.  Thin host code over
// the building blocks (K1 gpk_gram, K2 gpk_potrf, K3 gpk_potrs[_inv], K4 gpk_predict_mean[_mfma], K5
// gpk_predict_var_inv[_split], K6 gpk_lml_terms / gpk_wtw / gpk_lml_grad): the target normalisation
// (sklearn/gaussian_process/_gpr.py:271-282), the K1 -> K2 -> K3 sequencing (_gpr.py:343-364), the query panel loop
// and the optimiser objective (_gpr.py:537-652) that the Python host side (device.py, gpr.py) otherwise provides.
// Host pointers in, host pointers out; the device buffers are owned by the handle.
#include <cmath>
#include <limits>

#include "gpk_internal.h"

struct gpk_model {
  int64_t N = 0, Np = 0;
  int D = 0, P = 0, n_ls = 0, normalize_y = 0;
  double sf2 = 1.0, noise = 0.0, jitter = 0.0, lml = 0.0;
  double ls[GPK_MAX_D_PREDICT] = {0}, ls_in[GPK_MAX_D_PREDICT] = {0}, center[GPK_MAX_D_PREDICT] = {0};
  double y_mean[GPK_MAX_P] = {0}, y_std[GPK_MAX_P] = {0};
  bool fitted = false, mfma_mean_ok = false;
  // fp64 state
  double *X = nullptr, *Yn = nullptr, *K = nullptr, *winv = nullptr, *W = nullptr, *alpha = nullptr;
  bool has_W = false;
  // fp32 serving copies (built on the first fp32 predict)
  float *Xf = nullptr, *alphaf = nullptr;
  void* W3 = nullptr;            // fp16 x 2 split of W in fragment order (gpk_split2_rows) ...
  float* w_scales = nullptr;     // ... and the power of two each 128-row block was scaled by
  float* w_absmax = nullptr;     // max |(float)W_ij| per 128-row block, left by gpk_trtri_absmax (the split's first pass)
  int f32_mean_ok = -1;          // fp32 serving gate on the mean (-1: not evaluated for the current alpha)
  // scratch of gpk_lml: a second factorisation that leaves the fitted one alone
  double *sK = nullptr, *sW = nullptr, *sKinv = nullptr, *sT = nullptr, *swinv = nullptr, *salpha = nullptr;
  // query staging
  void *q = nullptr, *mean = nullptr, *work = nullptr, *work3 = nullptr, *q64 = nullptr;
  double* var = nullptr;
  size_t q_bytes = 0, mean_bytes = 0, work_bytes = 0, work3_bytes = 0, var_bytes = 0, q64_bytes = 0;
};

namespace {

void free_all(gpk_model* m) {
  void* ptrs[] = {m->X, m->Yn, m->K, m->winv, m->W, m->alpha, m->Xf, m->alphaf, m->W3, m->w_scales, m->w_absmax, m->sK, m->sW, m->sKinv, m->sT,
                  m->swinv, m->salpha, m->q, m->mean, m->work, m->work3, m->q64, m->var};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
}

template <typename T>
int dev_alloc(gpk_handle h, T** p, size_t count) {
  GPK_CHECK_HIP(h, hipMalloc((void**)p, count * sizeof(T)));
  if (h->debug_fill) GPK_CHECK_HIP(h, hipMemsetAsync(*p, 0xFF, count * sizeof(T), h->stream));
  return GPK_OK;
}

int grow(gpk_handle h, void** p, size_t* have, size_t need) {
  if (need <= *have) {
    if (h->debug_fill && need) GPK_CHECK_HIP(h, hipMemsetAsync(*p, 0xFF, need, h->stream));
    return GPK_OK;
  }
  GPK_CHECK_HIP(h, hipStreamSynchronize(h->stream));
  if (*p) GPK_CHECK_HIP(h, hipFree(*p));
  *p = nullptr; *have = 0;
  GPK_CHECK_HIP(h, hipMalloc(p, need));
  *have = need;
  if (h->debug_fill) GPK_CHECK_HIP(h, hipMemsetAsync(*p, 0xFF, need, h->stream));
  return GPK_OK;
}

__global__ void to_float_kernel(const double* __restrict__ s, long long n, float* __restrict__ d) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) d[i] = (float)s[i];
}

// var_n[m] * y_std[p]^2 -> out[m][p]  (sklearn/_gpr.py:487-489: undo the normalisation of the variance)
template <typename T>
__global__ void scale_var_kernel(const double* __restrict__ var, long long M, int P, const double* __restrict__ ystd,
                                 T* __restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < M * P) { const double s = ystd[i % P]; out[i] = (T)(var[i / P] * s * s); }
}

// the fp32 matrix-core mean kernel expands |a - b|^2 around `center`; admissible while the largest scaled squared
// norm (exp2 units) stays below 64 (device.py MFMA_MEAN_R2_MAX; gpk.h gpk_predict_mean_mfma)
bool mfma_mean_admissible(const double* X, int64_t N, int D, int P, const double* ls, double* center) {
  for (int d = 0; d < D; ++d) {
    double s = 0.0;
    for (int64_t i = 0; i < N; ++i) s += X[i * D + d];
    center[d] = s / (double)N;
  }
  double r2 = 0.0;
  for (int64_t i = 0; i < N; ++i) {
    double s = 0.0;
    for (int d = 0; d < D; ++d) { const double u = (X[i * D + d] - center[d]) / ls[d]; s += u * u; }
    if (s > r2) r2 = s;
  }
  return P <= 8 && 0.5 * 1.4426950408889634 * r2 <= 64.0;
}

// rows i_s = round(s (N - 1) / (S - 1)) of X (N x D) -> q (S x D)
__global__ void gather_rows_kernel(const double* __restrict__ X, long long N, int D, int S, double* __restrict__ q) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= S * D) return;
  const int s = e / D, d = e - s * D;
  const long long i = S > 1 ? (long long)llrint((double)s * (double)(N - 1) / (double)(S - 1)) : 0;
  q[e] = X[i * D + d];
}
__global__ void square_kernel(const double* __restrict__ a, long long n, double* __restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = a[i] * a[i];
}

// ---- fp32 serving gates (the rule of DeviceGP.predict_gated_dev, device.py) ------------------------------------------
// An fp32 kernel value carries the rounding of its exponent, so every term k*_j alpha_j of the mean is off by a few 1e-7
// of itself with random sign: mean error ~ c sqrt(sum_j (k*_j alpha_j)^2), c = 4.0e-7 for the matrix-core kernel and
// 9.0e-7 for the exact-difference kernel (worst query per batch over 41 random models: profiles/r02_fp32_gate_calibration.log).
// A2 = max_m sqrt(sum_j (k_mj alpha_j)^2) / max_m |mean_m| is measured on <= 1024 evenly spaced training rows (where it is
// largest) with two fp64 K4 launches - the second with the squared kernel (length-scales / sqrt 2, sf2^2) and squared
// weights - once per alpha; a model with c A2 above 1e-4 is served by the fp64 kernels.
// (round-4 calibration on batches up to 2^20 queries: profiles/r04_fp32_gate_calibration.log, device.py FP32_MEAN_ERR_PER_AMP)
constexpr double F32_MEAN_C_MFMA = 7.0e-7, F32_MEAN_C_VALU = 1.1e-6, F32_MEAN_TOL = 1e-4;
// fp32 variances below this fraction of the prior variance are recomputed in fp64 (their relative error is the absolute
// error of |W k*|^2 - up to 4e-5 kss - over the variance itself)
constexpr double F32_VAR_RECHECK_FRACTION = 1e-2;

int f32_mean_gate(gpk_handle h, gpk_model* m, bool* ok) {
  if (m->f32_mean_ok >= 0) { *ok = m->f32_mean_ok != 0; return GPK_OK; }
  const int D = m->D, P = m->P, S = (int)(m->N < 1024 ? m->N : 1024);
  double *q = nullptr, *a2 = nullptr, *o1 = nullptr, *o2 = nullptr;
  int rc = GPK_OK;
  if (hipMalloc((void**)&q, (size_t)S * D * sizeof(double)) != hipSuccess ||
      hipMalloc((void**)&a2, (size_t)m->N * P * sizeof(double)) != hipSuccess ||
      hipMalloc((void**)&o1, (size_t)S * P * sizeof(double)) != hipSuccess ||
      hipMalloc((void**)&o2, (size_t)S * P * sizeof(double)) != hipSuccess) {
    h->err = "fp32 mean gate: hipMalloc failed";
    rc = GPK_HIP_ERROR;
  }
  std::vector<double> h1((size_t)S * P), h2((size_t)S * P);
  if (rc == GPK_OK) {
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((S * D + 255) / 256)), dim3(256), 0, h->stream, m->X,
                       (long long)m->N, D, S, q);
    const long long na = m->N * P;
    hipLaunchKernelGGL(square_kernel, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, h->stream, m->alpha, na, a2);
    double zeros[GPK_MAX_P] = {0}, ones[GPK_MAX_P], ls2[GPK_MAX_D_PREDICT];
    for (int p = 0; p < GPK_MAX_P; ++p) ones[p] = 1.0;
    for (int d = 0; d < D; ++d) ls2[d] = m->ls[d] / std::sqrt(2.0);
    rc = gpk_predict_mean(h, GPK_F64, m->X, m->alpha, m->N, D, P, m->ls, m->sf2, zeros, ones, q, S, o1);
    if (rc == GPK_OK) rc = gpk_predict_mean(h, GPK_F64, m->X, a2, m->N, D, P, ls2, m->sf2 * m->sf2, zeros, ones, q, S, o2);
    if (rc == GPK_OK &&
        (hipMemcpyAsync(h1.data(), o1, h1.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
         hipMemcpyAsync(h2.data(), o2, h2.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
         hipStreamSynchronize(h->stream) != hipSuccess)) {
      h->err = "fp32 mean gate: copy failed";
      rc = GPK_HIP_ERROR;
    }
  }
  for (void* ptr : {(void*)q, (void*)a2, (void*)o1, (void*)o2})
    if (ptr) (void)hipFree(ptr);
  GPK_TRY(rc);
  double amp = 0.0;
  for (int p = 0; p < P; ++p) {
    double b = 0.0, a = 0.0;
    for (int s2 = 0; s2 < S; ++s2) {
      b = std::fmax(b, std::fabs(h1[(size_t)s2 * P + p]));
      a = std::fmax(a, h2[(size_t)s2 * P + p]);
    }
    amp = std::fmax(amp, std::sqrt(a) / std::fmax(b, 1e-300));
  }
  m->f32_mean_ok = ((m->mfma_mean_ok ? F32_MEAN_C_MFMA : F32_MEAN_C_VALU) * amp <= F32_MEAN_TOL) ? 1 : 0;
  *ok = m->f32_mean_ok != 0;
  return GPK_OK;
}

int set_hyper(gpk_handle h, gpk_model* m, const double* ls, int n_ls, double sf2, double noise, double jitter) {
  GPK_REQUIRE(h, ls && (n_ls == 1 || n_ls == m->D), "length-scales: n_ls must be 1 (isotropic) or D (ARD)");
  GPK_REQUIRE(h, sf2 > 0.0 && noise >= 0.0 && jitter >= 0.0, "sf2 must be positive, noise and jitter non-negative");
  for (int d = 0; d < m->D; ++d) {
    m->ls[d] = ls[n_ls == 1 ? 0 : d];
    GPK_REQUIRE(h, m->ls[d] > 0.0 && std::isfinite(m->ls[d]), "length-scales must be positive");
  }
  for (int d = 0; d < n_ls; ++d) m->ls_in[d] = ls[d];
  m->n_ls = n_ls; m->sf2 = sf2; m->noise = noise; m->jitter = jitter;
  return GPK_OK;
}

// W = L^-1 (one-off N^3/3 flops) - every later variance call and the alpha solve are then single GEMM launches
int ensure_W(gpk_handle h, gpk_model* m) {
  if (m->has_W) return GPK_OK;
  if (!m->W) GPK_TRY(dev_alloc(h, &m->W, (size_t)m->Np * m->Np));
  double* T = nullptr;
  const size_t tsz = (size_t)(m->Np / 2 + 128) * (m->Np / 2 + 128);
  GPK_CHECK_HIP(h, hipMalloc((void**)&T, tsz * sizeof(double)));
  int rc = GPK_OK;
  if (!m->w_absmax && hipMalloc((void**)&m->w_absmax, (size_t)(m->Np / 128) * sizeof(float)) != hipSuccess) {
    rc = GPK_HIP_ERROR;
    h->err = "fit: out of device memory";
  }
  if (rc == GPK_OK) rc = gpk_trtri_absmax(h, m->K, m->Np, m->Np, m->winv, m->W, m->Np, T, m->w_absmax);
  if (rc == GPK_OK && hipStreamSynchronize(h->stream) != hipSuccess) rc = GPK_HIP_ERROR;
  (void)hipFree(T);
  GPK_TRY(rc);
  m->has_W = true;
  if (m->W3) { (void)hipFree(m->W3); m->W3 = nullptr; }
  return GPK_OK;
}

int new_model(gpk_handle h, int64_t N, int D, int P, gpk_model** out) {
  GPK_REQUIRE(h, N >= 1 && D >= 1 && D <= GPK_MAX_D_PREDICT && P >= 1 && P <= GPK_MAX_P,
              "need N >= 1, 1 <= D <= GPK_MAX_D_PREDICT, 1 <= P <= GPK_MAX_P");
  GPK_REQUIRE(h, h->batch == 1, "composite calls are not available in batched mode");
  GPK_CHECK_HIP(h, hipSetDevice(h->device));
  GPK_CHECK_HIP(h, hipStreamSynchronize(h->stream));
  if (h->model) { free_all(h->model); delete h->model; h->model = nullptr; }
  gpk_model* m = new gpk_model();
  h->model = m;
  m->N = N; m->Np = gpk_padded(N); m->D = D; m->P = P;
  GPK_TRY(dev_alloc(h, &m->X, (size_t)N * D));
  GPK_TRY(dev_alloc(h, &m->Yn, (size_t)N * P));
  GPK_TRY(dev_alloc(h, &m->alpha, (size_t)N * P));
  GPK_TRY(dev_alloc(h, &m->K, (size_t)m->Np * m->Np));
  GPK_TRY(dev_alloc(h, &m->winv, (size_t)m->Np * GPK_TILE));
  *out = m;
  return GPK_OK;
}

// up to this padded size W is formed at fit time (cheap); larger models form it on the first variance request
constexpr int64_t EAGER_W_NP = 32768;

}  // namespace

void gpk_bmodel_free(gpk_handle h);

void gpk_model_free(gpk_handle h) {
  if (h->model) { free_all(h->model); delete h->model; h->model = nullptr; }
  gpk_bmodel_free(h);
}

extern "C" int gpk_model_release(gpk_handle h) {
  if (!h) return GPK_BAD_ARG;
  GPK_CHECK_HIP(h, hipSetDevice(h->device));
  GPK_CHECK_HIP(h, hipStreamSynchronize(h->stream));
  gpk_model_free(h);
  return GPK_OK;
}

extern "C" int gpk_fit(gpk_handle h, const double* X, int64_t N, int D, const double* Y, int P, const double* ls,
                       int n_ls, double sf2, double noise, double jitter, int normalize_y) {
  if (!h) return GPK_BAD_ARG;
  GPK_REQUIRE(h, X && Y, "fit: null pointer");
  gpk_model* m = nullptr;
  GPK_TRY(new_model(h, N, D, P, &m));
  GPK_TRY(set_hyper(h, m, ls, n_ls, sf2, noise, jitter));
  for (int64_t i = 0; i < N * D; ++i) GPK_REQUIRE(h, std::isfinite(X[i]), "fit: X contains NaN or infinity");
  // target normalisation: population std, a (numerically) zero std counts as 1   (_gpr.py:271-282)
  std::vector<double> yn((size_t)N * P);
  m->normalize_y = normalize_y ? 1 : 0;
  for (int p = 0; p < P; ++p) {
    double mean = 0.0, std_ = 1.0;
    for (int64_t i = 0; i < N; ++i) GPK_REQUIRE(h, std::isfinite(Y[i * P + p]), "fit: Y contains NaN or infinity");
    if (normalize_y) {
      for (int64_t i = 0; i < N; ++i) mean += Y[i * P + p];
      mean /= (double)N;
      double v = 0.0;
      for (int64_t i = 0; i < N; ++i) { const double d = Y[i * P + p] - mean; v += d * d; }
      std_ = std::sqrt(v / (double)N);
      if (std_ < 10.0 * std::numeric_limits<double>::epsilon()) std_ = 1.0;
    }
    m->y_mean[p] = mean; m->y_std[p] = std_;
    for (int64_t i = 0; i < N; ++i) yn[(size_t)i * P + p] = (Y[i * P + p] - mean) / std_;
  }
  m->mfma_mean_ok = mfma_mean_admissible(X, N, D, P, m->ls, m->center);
  GPK_CHECK_HIP(h, hipMemcpyAsync(m->X, X, (size_t)N * D * sizeof(double), hipMemcpyHostToDevice, h->stream));
  GPK_CHECK_HIP(h, hipMemcpyAsync(m->Yn, yn.data(), (size_t)N * P * sizeof(double), hipMemcpyHostToDevice, h->stream));
  GPK_CHECK_HIP(h, hipStreamSynchronize(h->stream));      // (yn leaves scope)
  // K1 -> K2 -> K3   (_gpr.py:343-364); a non-positive-definite matrix is reported as GPK_NOT_PD (_gpr.py:350-358)
  GPK_TRY(gpk_gram(h, GPK_F64, m->X, N, D, m->ls, sf2, noise + jitter, m->K, m->Np));
  int info = 0;
  GPK_TRY(gpk_potrf(h, m->K, m->Np, m->Np, m->winv, &info));
  if (m->Np <= EAGER_W_NP) {
    GPK_TRY(ensure_W(h, m));
    GPK_TRY(gpk_potrs_inv(h, m->W, m->Np, m->Np, m->Yn, N, P, m->alpha));
  } else {
    GPK_TRY(gpk_potrs(h, m->K, m->Np, m->Np, m->winv, m->Yn, N, P, m->alpha));
  }
  double terms[1 + GPK_MAX_P];
  GPK_TRY(gpk_lml_terms(h, m->K, N, m->Np, m->Yn, m->alpha, P, terms));
  m->lml = 0.0;
  for (int p = 0; p < P; ++p) m->lml += -0.5 * terms[1 + p] - terms[0] - 0.5 * (double)N * std::log(2.0 * M_PI);
  m->fitted = true;
  return GPK_OK;
}

extern "C" int gpk_predict(gpk_handle h, const void* Xq, int64_t M, void* mean, void* var, int dtype,
                           int var_includes_noise) {
  if (!h) return GPK_BAD_ARG;
  gpk_model* m = h->model;
  GPK_REQUIRE(h, m && m->fitted, "predict: no model (call gpk_fit or gpk_import first)");
  GPK_REQUIRE(h, Xq && mean && M >= 1, "predict: null pointer or empty batch");
  GPK_REQUIRE(h, dtype == GPK_F32 || dtype == GPK_F64, "predict: bad dtype");
  GPK_CHECK_HIP(h, hipSetDevice(h->device));
  const bool f32 = dtype == GPK_F32;
  const size_t es = f32 ? 4 : 8;
  const int D = m->D, P = m->P;
  for (int64_t i = 0; i < M * D; ++i) {
    const double v = f32 ? (double)((const float*)Xq)[i] : ((const double*)Xq)[i];
    GPK_REQUIRE(h, std::isfinite(v), "predict: Xq contains NaN or infinity");
  }
  // sklearn surface: k** = sf2 + noise (Sum.diag), clipped at 0; package surface: k** = sf2, floored at 1e-10
  const double kss = m->sf2 + (var_includes_noise ? m->noise : 0.0), floor_ = var_includes_noise ? 0.0 : 1e-10;
  if (var) GPK_TRY(ensure_W(h, m));
  if (f32) {
    // fp32 serving is gated: a model whose fp32 mean would leave the stated 1e-4 is served by the fp64 kernels
    bool ok = true;
    GPK_TRY(f32_mean_gate(h, m, &ok));
    if (!ok) {
      std::vector<double> q64((size_t)M * D), m64((size_t)M * P), v64(var ? (size_t)M * P : 0);
      for (int64_t i = 0; i < M * D; ++i) q64[(size_t)i] = (double)((const float*)Xq)[i];
      GPK_TRY(gpk_predict(h, q64.data(), M, m64.data(), var ? v64.data() : nullptr, GPK_F64, var_includes_noise));
      for (int64_t i = 0; i < M * P; ++i) ((float*)mean)[i] = (float)m64[(size_t)i];
      if (var)
        for (int64_t i = 0; i < M * P; ++i) ((float*)var)[i] = (float)v64[(size_t)i];
      return GPK_OK;
    }
  }
  // control-loop batches, fp64: the one-call serving path (two launches up to 32 rows)
  if (!f32 && M <= 64 && (!var || m->Np <= GPK_SMALL_MAX_NP)) {
    std::vector<double> v1((size_t)M);
    GPK_TRY(gpk_predict_host(h, m->X, m->alpha, m->N, D, P, m->ls, m->sf2, m->y_mean, m->y_std, var ? m->W : nullptr,
                             m->Np, m->Np, kss, floor_, (const double*)Xq, M, (double*)mean, var ? v1.data() : nullptr));
    if (var)
      for (int64_t i = 0; i < M; ++i)
        for (int p = 0; p < P; ++p) ((double*)var)[i * P + p] = v1[(size_t)i] * m->y_std[p] * m->y_std[p];
    return GPK_OK;
  }
  if (f32 && !m->Xf) {
    GPK_TRY(dev_alloc(h, &m->Xf, (size_t)m->N * D));
    GPK_TRY(dev_alloc(h, &m->alphaf, (size_t)m->N * P));
    const long long nx = m->N * D, na = m->N * P;
    hipLaunchKernelGGL(to_float_kernel, dim3((unsigned)((nx + 255) / 256)), dim3(256), 0, h->stream, m->X, nx, m->Xf);
    hipLaunchKernelGGL(to_float_kernel, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, h->stream, m->alpha, na, m->alphaf);
    GPK_LAUNCH_CHECK(h);
  }
  if (f32 && var && !m->W3) {
    // fp32 serving form of K5: W as two fp16 parts per entry, straight from the fp64 inverse factor (no fp32 copy).
    // All or nothing: a failure leaves no half-built operand behind for the next call to trust.
    int rc = (hipMalloc(&m->W3, (size_t)m->Np * m->Np * 4) == hipSuccess &&
              hipMalloc((void**)&m->w_scales, (size_t)(m->Np / 128) * sizeof(float)) == hipSuccess) ? GPK_OK : GPK_HIP_ERROR;
    if (rc != GPK_OK) h->err = "predict: out of device memory for the split inverse factor";
    // (the block maxima came out of gpk_trtri_absmax's epilogues: one pass over W)
    if (rc == GPK_OK && hipMemcpyAsync(m->w_scales, m->w_absmax, (size_t)(m->Np / 128) * sizeof(float), hipMemcpyDeviceToDevice,
                                       h->stream) != hipSuccess) rc = GPK_HIP_ERROR;
    if (rc == GPK_OK) rc = gpk_split2_rows_f64_absmax(h, m->W, m->Np, m->Np, m->w_scales, m->W3);
    if (rc == GPK_OK && hipStreamSynchronize(h->stream) != hipSuccess) { rc = GPK_HIP_ERROR; h->err = "predict: split of the inverse factor failed"; }
    if (rc != GPK_OK) {
      if (m->W3) { (void)hipFree(m->W3); m->W3 = nullptr; }
      if (m->w_scales) { (void)hipFree(m->w_scales); m->w_scales = nullptr; }
      return rc;
    }
  }
  // panel loop: <= 16384 queries and <= 4 GiB of K* per panel
  int64_t panel = (int64_t)((4ull << 30) / ((size_t)m->Np * es)) / GPK_TILE * GPK_TILE;
  if (panel > 16384) panel = 16384;
  if (panel < GPK_TILE) panel = GPK_TILE;
  if (panel > gpk_padded(M)) panel = gpk_padded(M);
  GPK_TRY(grow(h, &m->q, &m->q_bytes, (size_t)panel * D * es));
  GPK_TRY(grow(h, &m->mean, &m->mean_bytes, (size_t)panel * P * es));
  if (var) {
    if (f32) GPK_TRY(grow(h, &m->work3, &m->work3_bytes, (size_t)m->Np * panel * 4));     // K* in split form only
    else GPK_TRY(grow(h, &m->work, &m->work_bytes, (size_t)m->Np * panel * es));
    GPK_TRY(grow(h, (void**)&m->var, &m->var_bytes, (size_t)panel * sizeof(double) + (size_t)panel * P * es + GPK_MAX_P * sizeof(double)));
  }
  double* d_ystd = nullptr;
  void* d_varout = nullptr;
  if (var) {
    d_ystd = m->var + panel;
    d_varout = (void*)(d_ystd + GPK_MAX_P);
    GPK_CHECK_HIP(h, hipMemcpyAsync(d_ystd, m->y_std, P * sizeof(double), hipMemcpyHostToDevice, h->stream));
  }
  for (int64_t m0 = 0; m0 < M; m0 += panel) {
    const int64_t mc = M - m0 < panel ? M - m0 : panel;
    GPK_CHECK_HIP(h, hipMemcpyAsync(m->q, (const char*)Xq + (size_t)m0 * D * es, (size_t)mc * D * es, hipMemcpyHostToDevice, h->stream));
    if (f32 && m->mfma_mean_ok)
      GPK_TRY(gpk_predict_mean_mfma(h, m->Xf, m->alphaf, m->N, D, P, m->ls, m->sf2, m->center, m->y_mean, m->y_std,
                                    (const float*)m->q, mc, (float*)m->mean));
    else
      GPK_TRY(gpk_predict_mean(h, dtype, f32 ? (const void*)m->Xf : (const void*)m->X,
                               f32 ? (const void*)m->alphaf : (const void*)m->alpha, m->N, D, P, m->ls, m->sf2, m->y_mean,
                               m->y_std, m->q, mc, m->mean));
    GPK_CHECK_HIP(h, hipMemcpyAsync((char*)mean + (size_t)m0 * P * es, m->mean, (size_t)mc * P * es, hipMemcpyDeviceToHost, h->stream));
    if (var) {
      if (f32)
        GPK_TRY(gpk_predict_var_inv_split2(h, m->Xf, m->N, D, m->ls, m->sf2, m->W3, m->w_scales, m->Np, (const float*)m->q, mc,
                                           kss, floor_, m->work3, m->var));
      else
        GPK_TRY(gpk_predict_var_inv(h, GPK_F64, m->X, m->N, D, m->ls, m->sf2, m->W, m->Np, m->Np, m->q, mc, kss, floor_,
                                    m->work, m->var));
      if (f32) {
        // fp32 variances that are a small fraction of the prior's are recomputed by the fp64 launch (F32_VAR_RECHECK_FRACTION)
        std::vector<double> vh((size_t)mc);
        GPK_CHECK_HIP(h, hipMemcpyAsync(vh.data(), m->var, (size_t)mc * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        GPK_CHECK_HIP(h, hipStreamSynchronize(h->stream));
        std::vector<int64_t> low;
        for (int64_t i = 0; i < mc; ++i)
          if (vh[(size_t)i] < F32_VAR_RECHECK_FRACTION * kss) low.push_back(i);
        for (size_t l0 = 0; l0 < low.size(); l0 += (size_t)panel) {
          const int64_t lc = (int64_t)std::min(low.size() - l0, (size_t)panel);
          std::vector<double> q64((size_t)lc * D), v2((size_t)lc);
          for (int64_t i = 0; i < lc; ++i)
            for (int d = 0; d < D; ++d) q64[(size_t)i * D + d] = (double)((const float*)Xq)[(size_t)(m0 + low[l0 + i]) * D + d];
          GPK_TRY(grow(h, &m->work, &m->work_bytes, (size_t)m->Np * gpk_padded(lc) * sizeof(double)));
          GPK_TRY(grow(h, &m->q64, &m->q64_bytes, (size_t)lc * D * sizeof(double) + (size_t)gpk_padded(lc) * sizeof(double)));
          double* dq = (double*)m->q64;
          double* dv = dq + (size_t)lc * D;
          GPK_CHECK_HIP(h, hipMemcpyAsync(dq, q64.data(), q64.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
          GPK_TRY(gpk_predict_var_inv(h, GPK_F64, m->X, m->N, D, m->ls, m->sf2, m->W, m->Np, m->Np, dq, lc, kss, floor_,
                                      m->work, dv));
          GPK_CHECK_HIP(h, hipMemcpyAsync(v2.data(), dv, (size_t)lc * sizeof(double), hipMemcpyDeviceToHost, h->stream));
          GPK_CHECK_HIP(h, hipStreamSynchronize(h->stream));
          for (int64_t i = 0; i < lc; ++i) vh[(size_t)low[l0 + i]] = v2[(size_t)i];
        }
        for (int64_t i = 0; i < mc; ++i)      // undo the normalisation of the variance (sklearn/_gpr.py:487-489)
          for (int p = 0; p < P; ++p)
            ((float*)var)[(size_t)(m0 + i) * P + p] = (float)(vh[(size_t)i] * m->y_std[p] * m->y_std[p]);
      } else {
        const long long tot = mc * P;
        hipLaunchKernelGGL(scale_var_kernel<double>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, m->var,
                           (long long)mc, P, d_ystd, (double*)d_varout);
        GPK_LAUNCH_CHECK(h);
        GPK_CHECK_HIP(h, hipMemcpyAsync((char*)var + (size_t)m0 * P * es, d_varout, (size_t)mc * P * es, hipMemcpyDeviceToHost, h->stream));
      }
    }
    GPK_CHECK_HIP(h, hipStreamSynchronize(h->stream));     // the staging blocks are reused by the next panel
  }
  return GPK_OK;
}

extern "C" int gpk_lml(gpk_handle h, const double* theta, int n_theta, double* lml, double* grad) {
  if (!h) return GPK_BAD_ARG;
  gpk_model* m = h->model;
  GPK_REQUIRE(h, m && m->fitted, "lml: no model (call gpk_fit or gpk_import first)");
  GPK_REQUIRE(h, lml, "lml: null pointer");
  if (!theta) {                      // the fitted model's own value (_gpr.py:560-563)
    GPK_REQUIRE(h, !grad, "lml: the gradient needs theta");
    *lml = m->lml;
    return GPK_OK;
  }
  GPK_REQUIRE(h, m->normalize_y != -1, "lml(theta): an imported model does not carry the training targets");
  // theta = log [ls (1 value: isotropic, or D values: ARD), noise]; sf2 and jitter stay as fitted
  GPK_REQUIRE(h, n_theta == 2 || n_theta == m->D + 1, "lml: theta must hold log length-scale(s) and log noise");
  GPK_CHECK_HIP(h, hipSetDevice(h->device));
  const int nl = n_theta - 1, D = m->D, P = m->P;
  double ls[GPK_MAX_D_PREDICT];
  for (int d = 0; d < D; ++d) ls[d] = std::exp(theta[nl == 1 ? 0 : d]);
  const double noise = std::exp(theta[nl]);
  const size_t nn = (size_t)m->Np * m->Np;
  if (!m->sK) {
    GPK_TRY(dev_alloc(h, &m->sK, nn));
    GPK_TRY(dev_alloc(h, &m->sW, nn));
    GPK_TRY(dev_alloc(h, &m->sT, (size_t)(m->Np / 2 + 128) * (m->Np / 2 + 128)));
    GPK_TRY(dev_alloc(h, &m->swinv, (size_t)m->Np * GPK_TILE));
    GPK_TRY(dev_alloc(h, &m->salpha, (size_t)m->N * P));
  }
  if (grad && !m->sKinv) GPK_TRY(dev_alloc(h, &m->sKinv, nn));
  // the whole evaluation as one chain of launches with one synchronisation (gpk_lml_eval)
  int info = 0;
  double terms[1 + GPK_MAX_P], g[GPK_MAX_D_PREDICT + 2];
  const int rc = gpk_lml_eval(h, m->X, m->N, D, ls, m->sf2, noise + m->jitter, noise, m->Yn, P, m->sK, m->Np, m->swinv, m->sW,
                              m->sT, m->salpha, grad ? m->sKinv : nullptr, terms, grad ? g : nullptr, &info);
  if (rc == GPK_NOT_PD) {            // inside an optimiser: LML = -inf, zero gradient (_gpr.py:586-589)
    *lml = -std::numeric_limits<double>::infinity();
    if (grad) for (int i = 0; i < n_theta; ++i) grad[i] = 0.0;
    return GPK_OK;
  }
  GPK_TRY(rc);
  double v = 0.0;
  for (int p = 0; p < P; ++p) v += -0.5 * terms[1 + p] - terms[0] - 0.5 * (double)m->N * std::log(2.0 * M_PI);
  *lml = v;
  if (grad) {
    if (nl == 1) { double s = 0.0; for (int d = 0; d < D; ++d) s += g[d]; grad[0] = s; }   // kernels.py:1574-1576
    else for (int d = 0; d < D; ++d) grad[d] = g[d];
    grad[nl] = g[D];
  }
  return GPK_OK;
}

extern "C" int gpk_export(gpk_handle h, int64_t* N, int* D, int* P, double* L, double* alpha, double* y_mean,
                          double* y_std, double* lml) {
  if (!h) return GPK_BAD_ARG;
  gpk_model* m = h->model;
  GPK_REQUIRE(h, m && m->fitted, "export: no model");
  GPK_CHECK_HIP(h, hipSetDevice(h->device));
  if (N) *N = m->N;
  if (D) *D = m->D;
  if (P) *P = m->P;
  if (L) {      // the lower factor, N x N row-major, zeros above the diagonal (what scikit-learn keeps as L_)
    GPK_CHECK_HIP(h, hipMemcpy2DAsync(L, (size_t)m->N * sizeof(double), m->K, (size_t)m->Np * sizeof(double),
                                      (size_t)m->N * sizeof(double), (size_t)m->N, hipMemcpyDeviceToHost, h->stream));
    GPK_CHECK_HIP(h, hipStreamSynchronize(h->stream));
    for (int64_t i = 0; i < m->N; ++i)
      for (int64_t j = i + 1; j < m->N; ++j) L[i * m->N + j] = 0.0;
  }
  if (alpha) GPK_CHECK_HIP(h, hipMemcpyAsync(alpha, m->alpha, (size_t)m->N * m->P * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  GPK_CHECK_HIP(h, hipStreamSynchronize(h->stream));
  for (int p = 0; p < m->P; ++p) {
    if (y_mean) y_mean[p] = m->y_mean[p];
    if (y_std) y_std[p] = m->y_std[p];
  }
  if (lml) *lml = m->lml;
  return GPK_OK;
}

extern "C" int gpk_import(gpk_handle h, const double* X, int64_t N, int D, const double* L, const double* alpha, int P,
                          const double* ls, int n_ls, double sf2, double noise, const double* y_mean, const double* y_std) {
  if (!h) return GPK_BAD_ARG;
  GPK_REQUIRE(h, X && L && alpha && y_mean && y_std, "import: null pointer");
  gpk_model* m = nullptr;
  GPK_TRY(new_model(h, N, D, P, &m));
  GPK_TRY(set_hyper(h, m, ls, n_ls, sf2, noise, 0.0));
  for (int p = 0; p < P; ++p) { m->y_mean[p] = y_mean[p]; m->y_std[p] = y_std[p]; }
  m->mfma_mean_ok = mfma_mean_admissible(X, N, D, P, m->ls, m->center);
  GPK_CHECK_HIP(h, hipMemcpyAsync(m->X, X, (size_t)N * D * sizeof(double), hipMemcpyHostToDevice, h->stream));
  GPK_CHECK_HIP(h, hipMemcpyAsync(m->alpha, alpha, (size_t)N * P * sizeof(double), hipMemcpyHostToDevice, h->stream));
  // the padded factor [L 0; 0 I]; Yn is not part of a stored model: Yn = (L L^T) alpha is not needed for prediction
  // (gpk_lml on an imported model is refused below)
  GPK_CHECK_HIP(h, hipMemsetAsync(m->K, 0, (size_t)m->Np * m->Np * sizeof(double), h->stream));
  GPK_CHECK_HIP(h, hipMemcpy2DAsync(m->K, (size_t)m->Np * sizeof(double), L, (size_t)N * sizeof(double),
                                    (size_t)N * sizeof(double), (size_t)N, hipMemcpyHostToDevice, h->stream));
  if (m->Np > N) {
    std::vector<double> ones((size_t)(m->Np - N), 1.0);
    GPK_CHECK_HIP(h, hipMemcpy2DAsync(m->K + N * m->Np + N, (size_t)(m->Np + 1) * sizeof(double), ones.data(), sizeof(double),
                                      sizeof(double), (size_t)(m->Np - N), hipMemcpyHostToDevice, h->stream));
    GPK_CHECK_HIP(h, hipStreamSynchronize(h->stream));
  }
  GPK_TRY(gpk_leaf_inverses(h, m->K, m->Np, m->Np, m->winv));
  if (m->Np <= EAGER_W_NP) GPK_TRY(ensure_W(h, m));
  // Yn = K alpha = L (L^T alpha) is recovered lazily only if gpk_lml is asked for: not supported on imported models
  GPK_CHECK_HIP(h, hipMemsetAsync(m->Yn, 0, (size_t)N * P * sizeof(double), h->stream));
  GPK_CHECK_HIP(h, hipStreamSynchronize(h->stream));
  m->lml = std::numeric_limits<double>::quiet_NaN();
  m->normalize_y = -1;               // unknown: marks an imported model (gpk_lml(theta) needs the training targets)
  m->fitted = true;
  return GPK_OK;
}

// ---- B single-output models on shared inputs (the per-axis GPs of src/px4/gp_trainer.py:139-179) ---------------------
// gpk_fit_batched / gpk_predict_batched / gpk_lml_batched: the B factorisations, inverse factors and alpha solves run
// as ONE launch chain (gpk_batch_begin: every kernel of the chain gets a batch grid dimension); the Gram build, the
// LML reductions and the gradient reduction run once per model (their hyper-parameters differ).
struct gpk_bmodel {
  int B = 0, D = 0, n_ls = 0, normalize_y = 0;
  int64_t N = 0, Ne = 0, Np = 0;
  double jitter = 0.0;
  double ls[GPK_MAX_BATCH][GPK_MAX_D_PREDICT] = {{0}};
  double sf2[GPK_MAX_BATCH] = {0}, noise[GPK_MAX_BATCH] = {0}, y_mean[GPK_MAX_BATCH] = {0}, y_std[GPK_MAX_BATCH] = {0};
  double lml[GPK_MAX_BATCH] = {0};
  bool fitted = false;
  double *X = nullptr, *Yn = nullptr, *alpha = nullptr, *alphaT = nullptr, *K = nullptr, *winv = nullptr, *W = nullptr;
  double *sK = nullptr, *sW = nullptr, *sKinv = nullptr, *sT = nullptr, *swinv = nullptr, *salpha = nullptr;   // gpk_lml_batched
  void *q = nullptr, *mean = nullptr, *work = nullptr;
  double* var = nullptr;
  size_t q_bytes = 0, mean_bytes = 0, work_bytes = 0, var_bytes = 0;
  size_t nn() const { return (size_t)Np * Np; }
  size_t tsz() const { return (size_t)(Np / 2 + 128) * (Np / 2 + 128); }
};

namespace {

// alphaT[i][b] = alpha[b][i]: gpk_predict_mean_multi wants one column per model
__global__ void rows_to_cols_kernel(const double* __restrict__ rows, long long N, long long Ne, int B, double* __restrict__ cols) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < N * B) cols[i] = rows[(i % B) * Ne + i / B];
}

// out[m][b] = var[m] * s2  (column b of the M x B variance block)
__global__ void scale_var_col_kernel(const double* __restrict__ var, long long M, int B, int b, double s2, double* __restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < M) out[i * B + b] = var[i] * s2;
}

// K2 + W = L^-1 + K3 (+ K^-1 = W^T W) for all models in one launch chain; info[b] != 0: model b is not positive definite
int batched_chain(gpk_handle h, gpk_bmodel* m, double* K, double* winv, double* W, double* T, double* alpha, double* Kinv,
                  int* info) {
  GPK_TRY(gpk_batch_begin(h, m->B));
  int rc = GPK_OK;
  const struct { const void* p; size_t stride; } bufs[] = {
      {K, m->nn() * 8}, {winv, (size_t)m->Np * GPK_TILE * 8}, {W, m->nn() * 8}, {T, m->tsz() * 8},
      {m->Yn, (size_t)m->Ne * 8}, {alpha, (size_t)m->Ne * 8}, {Kinv, m->nn() * 8}};
  for (const auto& b : bufs)
    if (rc == GPK_OK && b.p) rc = gpk_batch_buffer(h, b.p, (int64_t)b.stride);
  if (rc == GPK_OK) {
    rc = gpk_potrf(h, K, m->Np, m->Np, winv, info);
    if (rc == GPK_NOT_PD) rc = GPK_OK;            // per-model outcome is in info[]
  }
  if (rc == GPK_OK) rc = gpk_trtri(h, K, m->Np, m->Np, winv, W, m->Np, T);
  if (rc == GPK_OK) rc = gpk_potrs_inv(h, W, m->Np, m->Np, m->Yn, m->N, 1, alpha);
  if (rc == GPK_OK && Kinv) rc = gpk_wtw(h, W, m->Np, m->Np, Kinv, m->Np);
  (void)gpk_batch_end(h);
  return rc;
}

void bfree_all(gpk_bmodel* m) {
  void* ptrs[] = {m->X, m->Yn, m->alpha, m->alphaT, m->K, m->winv, m->W, m->sK, m->sW, m->sKinv, m->sT, m->swinv,
                  m->salpha, m->q, m->mean, m->work, m->var};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
}

}  // namespace

void gpk_bmodel_free(gpk_handle h) {
  if (h->bmodel) { bfree_all(h->bmodel); delete h->bmodel; h->bmodel = nullptr; }
}

extern "C" int gpk_fit_batched(gpk_handle h, int B, const double* X, int64_t N, int D, const double* Y, const double* ls,
                               int n_ls, const double* sf2, const double* noise, double jitter, int normalize_y, int* info) {
  if (!h) return GPK_BAD_ARG;
  GPK_REQUIRE(h, X && Y && ls && sf2 && noise && info, "fit_batched: null pointer");
  GPK_REQUIRE(h, B >= 1 && B <= GPK_MAX_BATCH, "fit_batched: 1..8 models");
  GPK_REQUIRE(h, N >= 1 && D >= 1 && D <= GPK_MAX_D_PREDICT, "fit_batched: need N >= 1 and 1 <= D <= GPK_MAX_D_PREDICT");
  GPK_REQUIRE(h, n_ls == 1 || n_ls == D, "fit_batched: n_ls must be 1 (isotropic) or D (ARD)");
  GPK_REQUIRE(h, jitter >= 0.0, "fit_batched: jitter must be non-negative");
  GPK_REQUIRE(h, h->batch == 1, "composite calls are not available in batched mode");
  for (int b = 0; b < B; ++b) {
    GPK_REQUIRE(h, sf2[b] > 0.0 && noise[b] >= 0.0, "fit_batched: sf2 must be positive, noise non-negative");
    for (int d = 0; d < n_ls; ++d)
      GPK_REQUIRE(h, ls[b * n_ls + d] > 0.0 && std::isfinite(ls[b * n_ls + d]), "fit_batched: length-scales must be positive");
  }
  for (int64_t i = 0; i < N * D; ++i) GPK_REQUIRE(h, std::isfinite(X[i]), "fit_batched: X contains NaN or infinity");
  for (int64_t i = 0; i < N * B; ++i) GPK_REQUIRE(h, std::isfinite(Y[i]), "fit_batched: Y contains NaN or infinity");
  GPK_CHECK_HIP(h, hipSetDevice(h->device));
  GPK_CHECK_HIP(h, hipStreamSynchronize(h->stream));
  gpk_bmodel_free(h);
  gpk_bmodel* m = new gpk_bmodel();
  h->bmodel = m;
  m->B = B; m->N = N; m->Ne = N + (N & 1); m->Np = gpk_padded(N); m->D = D; m->n_ls = n_ls; m->jitter = jitter;
  m->normalize_y = normalize_y ? 1 : 0;
  for (int b = 0; b < B; ++b) {
    for (int d = 0; d < D; ++d) m->ls[b][d] = ls[b * n_ls + (n_ls == 1 ? 0 : d)];
    m->sf2[b] = sf2[b]; m->noise[b] = noise[b];
  }
  // per-model rows (stride Ne: the batch strides must be multiples of 16 bytes, so odd N is padded by one entry)
  std::vector<double> yn((size_t)B * m->Ne, 0.0);
  for (int b = 0; b < B; ++b) {
    double mean = 0.0, std_ = 1.0;
    if (normalize_y) {          // population std; a (numerically) zero std counts as 1   (_gpr.py:271-282)
      for (int64_t i = 0; i < N; ++i) mean += Y[i * B + b];
      mean /= (double)N;
      double v = 0.0;
      for (int64_t i = 0; i < N; ++i) { const double d = Y[i * B + b] - mean; v += d * d; }
      std_ = std::sqrt(v / (double)N);
      if (std_ < 10.0 * std::numeric_limits<double>::epsilon()) std_ = 1.0;
    }
    m->y_mean[b] = mean; m->y_std[b] = std_;
    for (int64_t i = 0; i < N; ++i) yn[(size_t)b * m->Ne + i] = (Y[i * B + b] - mean) / std_;
  }
  GPK_TRY(dev_alloc(h, &m->X, (size_t)N * D));
  GPK_TRY(dev_alloc(h, &m->Yn, (size_t)B * m->Ne));
  GPK_TRY(dev_alloc(h, &m->alpha, (size_t)B * m->Ne));
  GPK_TRY(dev_alloc(h, &m->alphaT, (size_t)N * B));
  GPK_TRY(dev_alloc(h, &m->K, (size_t)B * m->nn()));
  GPK_TRY(dev_alloc(h, &m->winv, (size_t)B * m->Np * GPK_TILE));
  GPK_TRY(dev_alloc(h, &m->W, (size_t)B * m->nn()));
  double* T = nullptr;
  GPK_TRY(dev_alloc(h, &T, (size_t)B * m->tsz()));
  int rc = GPK_OK;
  if (hipMemcpyAsync(m->X, X, (size_t)N * D * sizeof(double), hipMemcpyHostToDevice, h->stream) != hipSuccess ||
      hipMemcpyAsync(m->Yn, yn.data(), yn.size() * sizeof(double), hipMemcpyHostToDevice, h->stream) != hipSuccess ||
      hipStreamSynchronize(h->stream) != hipSuccess)
    rc = GPK_HIP_ERROR;
  for (int b = 0; b < B && rc == GPK_OK; ++b)       // K1 per model
    rc = gpk_gram(h, GPK_F64, m->X, N, D, m->ls[b], m->sf2[b], m->noise[b] + jitter, m->K + (size_t)b * m->nn(), m->Np);
  if (rc == GPK_OK) rc = batched_chain(h, m, m->K, m->winv, m->W, T, m->alpha, nullptr, info);
  if (rc == GPK_OK && hipStreamSynchronize(h->stream) != hipSuccess) rc = GPK_HIP_ERROR;
  (void)hipFree(T);
  GPK_TRY(rc);
  bool all_pd = true;
  for (int b = 0; b < B; ++b) {
    if (info[b] != 0) { all_pd = false; m->lml[b] = -std::numeric_limits<double>::infinity(); continue; }
    double terms[2];
    GPK_TRY(gpk_lml_terms(h, m->K + (size_t)b * m->nn(), N, m->Np, m->Yn + (size_t)b * m->Ne, m->alpha + (size_t)b * m->Ne, 1, terms));
    m->lml[b] = -0.5 * terms[1] - terms[0] - 0.5 * (double)N * std::log(2.0 * M_PI);
  }
  if (!all_pd) {
    h->err = "fit_batched: a model's matrix is not positive definite (see info[])";
    return GPK_NOT_PD;
  }
  const long long tot = (long long)N * B;
  hipLaunchKernelGGL(rows_to_cols_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, m->alpha, (long long)N,
                     (long long)m->Ne, B, m->alphaT);
  GPK_LAUNCH_CHECK(h);
  GPK_CHECK_HIP(h, hipStreamSynchronize(h->stream));
  m->fitted = true;
  return GPK_OK;
}

extern "C" int gpk_predict_batched(gpk_handle h, const double* Xq, int64_t M, double* mean, double* var,
                                   int var_includes_noise) {
  if (!h) return GPK_BAD_ARG;
  gpk_bmodel* m = h->bmodel;
  GPK_REQUIRE(h, m && m->fitted, "predict_batched: no model (call gpk_fit_batched first)");
  GPK_REQUIRE(h, Xq && mean && M >= 1, "predict_batched: null pointer or empty batch");
  GPK_CHECK_HIP(h, hipSetDevice(h->device));
  const int B = m->B, D = m->D;
  for (int64_t i = 0; i < M * D; ++i) GPK_REQUIRE(h, std::isfinite(Xq[i]), "predict_batched: Xq contains NaN or infinity");
  double kss[GPK_MAX_BATCH];
  for (int b = 0; b < B; ++b) kss[b] = m->sf2[b] + (var_includes_noise ? m->noise[b] : 0.0);
  const double floor_ = var_includes_noise ? 0.0 : 1e-10;
  double lsBD[GPK_MAX_BATCH * GPK_MAX_D_PREDICT];      // the length-scales as one contiguous (B x D) block
  for (int b = 0; b < B; ++b)
    for (int d = 0; d < D; ++d) lsBD[b * D + d] = m->ls[b][d];
  // control-loop batches: one call, two launches for all models (src/px4/pretrained_gp.py:52-98)
  if (M <= 32 && m->Np <= GPK_SMALL_MAX_NP) {
    const double *Xs[GPK_MAX_BATCH], *as[GPK_MAX_BATCH], *Ws[GPK_MAX_BATCH];
    for (int b = 0; b < B; ++b) { Xs[b] = m->X; as[b] = m->alpha + (size_t)b * m->Ne; Ws[b] = m->W + (size_t)b * m->nn(); }
    std::vector<double> mb((size_t)B * M), vb(var ? (size_t)B * M : 0);
    GPK_TRY(gpk_predict_host_multi(h, B, Xs, as, m->N, D, lsBD, m->sf2, m->y_mean, m->y_std, var ? Ws : nullptr, m->Np,
                                   m->Np, var ? kss : nullptr, floor_, Xq, M, mb.data(), var ? vb.data() : nullptr));
    for (int64_t i = 0; i < M; ++i)
      for (int b = 0; b < B; ++b) {
        mean[i * B + b] = mb[(size_t)b * M + i];
        if (var) var[i * B + b] = vb[(size_t)b * M + i] * m->y_std[b] * m->y_std[b];
      }
    return GPK_OK;
  }
  int64_t panel = (int64_t)((4ull << 30) / ((size_t)m->Np * 8)) / GPK_TILE * GPK_TILE;
  if (panel > 16384) panel = 16384;
  if (panel < GPK_TILE) panel = GPK_TILE;
  if (panel > gpk_padded(M)) panel = gpk_padded(M);
  GPK_TRY(grow(h, &m->q, &m->q_bytes, (size_t)panel * D * 8));
  GPK_TRY(grow(h, &m->mean, &m->mean_bytes, (size_t)panel * B * 8));
  if (var) {
    GPK_TRY(grow(h, &m->work, &m->work_bytes, (size_t)m->Np * panel * 8));
    GPK_TRY(grow(h, (void**)&m->var, &m->var_bytes, (size_t)panel * 8 + (size_t)panel * B * 8));
  }
  double* d_varout = var ? m->var + panel : nullptr;
  for (int64_t m0 = 0; m0 < M; m0 += panel) {
    const int64_t mc = M - m0 < panel ? M - m0 : panel;
    GPK_CHECK_HIP(h, hipMemcpyAsync(m->q, Xq + (size_t)m0 * D, (size_t)mc * D * 8, hipMemcpyHostToDevice, h->stream));
    GPK_TRY(gpk_predict_mean_multi(h, GPK_F64, m->X, m->alphaT, m->N, D, B, lsBD, m->sf2, m->y_mean, m->y_std, m->q, mc, m->mean));
    GPK_CHECK_HIP(h, hipMemcpyAsync(mean + (size_t)m0 * B, m->mean, (size_t)mc * B * 8, hipMemcpyDeviceToHost, h->stream));
    if (var) {
      for (int b = 0; b < B; ++b) {          // K* differs per model (its own length-scales): one variance launch each
        GPK_TRY(gpk_predict_var_inv(h, GPK_F64, m->X, m->N, D, m->ls[b], m->sf2[b], m->W + (size_t)b * m->nn(), m->Np, m->Np,
                                    m->q, mc, kss[b], floor_, m->work, m->var));
        hipLaunchKernelGGL(scale_var_col_kernel, dim3((unsigned)((mc + 255) / 256)), dim3(256), 0, h->stream, m->var,
                           (long long)mc, B, b, m->y_std[b] * m->y_std[b], d_varout);
        GPK_LAUNCH_CHECK(h);
      }
      GPK_CHECK_HIP(h, hipMemcpyAsync(var + (size_t)m0 * B, d_varout, (size_t)mc * B * 8, hipMemcpyDeviceToHost, h->stream));
    }
    GPK_CHECK_HIP(h, hipStreamSynchronize(h->stream));
  }
  return GPK_OK;
}

extern "C" int gpk_lml_batched(gpk_handle h, const double* thetas, int n_theta, double* lml, double* grad) {
  if (!h) return GPK_BAD_ARG;
  gpk_bmodel* m = h->bmodel;
  GPK_REQUIRE(h, m && m->fitted, "lml_batched: no model (call gpk_fit_batched first)");
  GPK_REQUIRE(h, lml, "lml_batched: null pointer");
  const int B = m->B, D = m->D;
  if (!thetas) {
    GPK_REQUIRE(h, !grad, "lml_batched: the gradient needs thetas");
    for (int b = 0; b < B; ++b) lml[b] = m->lml[b];
    return GPK_OK;
  }
  GPK_REQUIRE(h, n_theta == 2 || n_theta == D + 1, "lml_batched: each theta row holds log length-scale(s) and log noise");
  GPK_CHECK_HIP(h, hipSetDevice(h->device));
  const int nl = n_theta - 1;
  double ls[GPK_MAX_BATCH][GPK_MAX_D_PREDICT], noise[GPK_MAX_BATCH];
  for (int b = 0; b < B; ++b) {
    for (int d = 0; d < D; ++d) ls[b][d] = std::exp(thetas[b * n_theta + (nl == 1 ? 0 : d)]);
    noise[b] = std::exp(thetas[b * n_theta + nl]);
  }
  if (!m->sK) {
    GPK_TRY(dev_alloc(h, &m->sK, (size_t)B * m->nn()));
    GPK_TRY(dev_alloc(h, &m->sW, (size_t)B * m->nn()));
    GPK_TRY(dev_alloc(h, &m->sT, (size_t)B * m->tsz()));
    GPK_TRY(dev_alloc(h, &m->swinv, (size_t)B * m->Np * GPK_TILE));
    GPK_TRY(dev_alloc(h, &m->salpha, (size_t)B * m->Ne));
  }
  if (grad && !m->sKinv) GPK_TRY(dev_alloc(h, &m->sKinv, (size_t)B * m->nn()));
  for (int b = 0; b < B; ++b)
    GPK_TRY(gpk_gram(h, GPK_F64, m->X, m->N, D, ls[b], m->sf2[b], noise[b] + m->jitter, m->sK + (size_t)b * m->nn(), m->Np));
  int info[GPK_MAX_BATCH] = {0};
  double terms[2 * GPK_MAX_BATCH], g[GPK_MAX_BATCH * (GPK_MAX_D_PREDICT + 2)], lsf[GPK_MAX_BATCH * GPK_MAX_D_PREDICT];
  for (int b = 0; b < B; ++b)
    for (int d = 0; d < D; ++d) lsf[b * D + d] = ls[b][d];
  // factor, inverse factor, alpha, K^-1, terms and gradient passes of all B models: one chain, one synchronisation
  GPK_TRY(gpk_lml_chain_batched(h, B, m->X, m->N, D, lsf, m->sf2, noise, m->Yn, m->Ne, m->sK, m->Np, m->swinv, m->sW, m->sT,
                                m->tsz(), m->salpha, grad ? m->sKinv : nullptr, terms, g, info));
  for (int b = 0; b < B; ++b) {
    if (info[b] != 0) {              // inside an optimiser: LML = -inf, zero gradient (_gpr.py:586-589)
      lml[b] = -std::numeric_limits<double>::infinity();
      if (grad) for (int i = 0; i < n_theta; ++i) grad[b * n_theta + i] = 0.0;
      continue;
    }
    lml[b] = -0.5 * terms[2 * b + 1] - terms[2 * b] - 0.5 * (double)m->N * std::log(2.0 * M_PI);
    if (grad) {
      const double* gs = g + (size_t)b * (D + 2);
      double* gb = grad + (size_t)b * n_theta;
      if (nl == 1) { double s = 0.0; for (int d = 0; d < D; ++d) s += gs[d]; gb[0] = s; }   // kernels.py:1574-1576
      else for (int d = 0; d < D; ++d) gb[d] = gs[d];
      gb[nl] = gs[D];
    }
  }
  return GPK_OK;
}
