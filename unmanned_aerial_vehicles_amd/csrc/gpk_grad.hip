// K6b: fused log-marginal-likelihood gradient reduction.
//
//   g_d     = 1/2 sum_ij Q_ij K_ij ((x_id - x_jd) / ls_d)^2          d = 0 .. D-1
//   g_noise = 1/2 noise sum_i Q_ii
//   g_sf2   = 1/2 sum_ij Q_ij K_ij                                   (K_ij = sf2 exp(-d2_ij / 2))
//   Q_ij    = sum_p alpha_ip alpha_jp - P Kinv_ij
//
// One streaming pass over the lower triangle of K^-1 (HBM-bound: N^2/2 doubles); K_ij and the
// per-feature factors are recomputed from X on the fly, so neither K nor the N x N x D
// gradient tensor of the reference is ever materialised.  64 x 64 tiles, 4 x 4 per thread;
// strictly-lower elements count twice (symmetry).  Block partials are reduced in a fixed
// order by a second kernel (deterministic).
#include "gpk_internal.h"
#include "gpk_math.h"

namespace {

constexpr int TS = 64, DMAXG = 16, GW = DMAXG + 2;   // block partial: D (<= 16) + noise + sf2
struct LsG { double v[DMAXG]; };

__device__ __forceinline__ double exp_neg64(double x) { return gpk_exp_neg(x); }

// grid-stride over lower-triangular 64 x 64 tiles; D <= 16
__global__ __launch_bounds__(256) void lml_grad_kernel(const double* __restrict__ X, long long N, int D, LsG ls,
                                                       double sf2, const double* __restrict__ alpha, int P,
                                                       const double* __restrict__ Kinv, long long ldk,
                                                       long long ntiles, double* __restrict__ partial) {
  __shared__ double xi[DMAXG * TS], xj[DMAXG * TS];
  __shared__ double ai[GPK_MAX_P * TS], aj[GPK_MAX_P * TS];
  __shared__ double red[4][GW];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  double g_noise = 0.0, g_sf2 = 0.0;
  double gd[DMAXG];
#pragma unroll
  for (int d = 0; d < DMAXG; ++d) gd[d] = 0.0;

  for (long long id = blockIdx.x; id < ntiles; id += gridDim.x) {
    long long ti = (long long)((__builtin_sqrt(8.0 * (double)id + 1.0) - 1.0) * 0.5);
    while ((ti + 1) * (ti + 2) / 2 <= id) ++ti;
    while (ti * (ti + 1) / 2 > id) --ti;
    const long long tj = id - ti * (ti + 1) / 2;
    const long long i0 = ti * TS, j0 = tj * TS;
    __syncthreads();
    for (int e = tid; e < TS * D; e += 256) {
      const int i = e / D, d = e - i * D;
      xi[d * TS + i] = (i0 + i < N) ? X[(i0 + i) * D + d] / ls.v[d] : 0.0;
      xj[d * TS + i] = (j0 + i < N) ? X[(j0 + i) * D + d] / ls.v[d] : 0.0;
    }
    for (int e = tid; e < TS * P; e += 256) {
      const int i = e / P, p = e - i * P;
      ai[p * TS + i] = (i0 + i < N) ? alpha[(i0 + i) * P + p] : 0.0;
      aj[p * TS + i] = (j0 + i < N) ? alpha[(j0 + i) * P + p] : 0.0;
    }
    __syncthreads();

    // pass 1: squared distances and alpha_i . alpha_j
    double d2[4][4], q[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) { d2[r][c] = 0.0; q[r][c] = 0.0; }
    for (int d = 0; d < D; ++d) {
      double a[4], b[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) a[r] = xi[d * TS + 4 * ty + r];
#pragma unroll
      for (int c = 0; c < 4; ++c) b[c] = xj[d * TS + 4 * tx + c];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) { const double df = a[r] - b[c]; d2[r][c] = __builtin_fma(df, df, d2[r][c]); }
    }
    for (int p = 0; p < P; ++p) {
      double a[4], b[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) a[r] = ai[p * TS + 4 * ty + r];
#pragma unroll
      for (int c = 0; c < 4; ++c) b[c] = aj[p * TS + 4 * tx + c];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) q[r][c] = __builtin_fma(a[r], b[c], q[r][c]);
    }
    // coefficient w_ij Q_ij K_ij  (w = 2 strictly below the diagonal, 1 on it, 0 above / padding)
    double coef[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long long gi = i0 + 4 * ty + r;
      const double2 k01 = *reinterpret_cast<const double2*>(Kinv + gi * ldk + j0 + 4 * tx);
      const double2 k23 = *reinterpret_cast<const double2*>(Kinv + gi * ldk + j0 + 4 * tx + 2);
      const double kin[4] = {k01.x, k01.y, k23.x, k23.y};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const long long gj = j0 + 4 * tx + c;
        const bool live = gi < N && gj < N && gj <= gi;
        const double w = live ? (gj == gi ? 1.0 : 2.0) : 0.0;
        const double Q = q[r][c] - (double)P * (live ? kin[c] : 0.0);
        const double kv = sf2 * exp_neg64(-0.5 * d2[r][c]);
        coef[r][c] = w * Q * kv;
        g_sf2 += coef[r][c];
        if (live && gj == gi) g_noise += Q;
      }
    }
    // pass 2: per-feature factors
#pragma unroll
    for (int d = 0; d < DMAXG; ++d) {
      if (d < D) {
        double a[4], b[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] = xi[d * TS + 4 * ty + r];
#pragma unroll
        for (int c = 0; c < 4; ++c) b[c] = xj[d * TS + 4 * tx + c];
        double s = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) { const double df = a[r] - b[c]; s = __builtin_fma(coef[r][c], df * df, s); }
        gd[d] += s;
      }
    }
  }

  // block reduction: wave shuffle, then 4 waves through LDS
  const int lane = tid & 63, wave = tid >> 6;
  auto wave_sum = [](double v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
  };
  __syncthreads();
#pragma unroll
  for (int d = 0; d < DMAXG; ++d) {
    const double v = wave_sum(gd[d]);
    if (lane == 0) red[wave][d] = v;
  }
  {
    const double vn = wave_sum(g_noise), vs = wave_sum(g_sf2);
    if (lane == 0) { red[wave][DMAXG] = vn; red[wave][DMAXG + 1] = vs; }
  }
  __syncthreads();
  if (tid < GW) partial[(long long)blockIdx.x * GW + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

__global__ __launch_bounds__(256) void grad_reduce_kernel(const double* __restrict__ partial, int nblocks,
                                                          double* __restrict__ out) {
  // one workgroup per output component; fixed-order strided sum + tree
  __shared__ double red[256];
  const int c = blockIdx.x, tid = threadIdx.x;
  double s = 0.0;
  for (int b = tid; b < nblocks; b += 256) s += partial[(long long)b * GW + c];
  red[tid] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) red[tid] += red[tid + off];
    __syncthreads();
  }
  if (tid == 0) out[c] = red[0];
}

}  // namespace

// the launches of gpk_lml_grad: the GW raw sums (before the factors of gpk_lml_grad's last lines) to dout (device)
int gpk_lml_grad_enqueue(gpk_handle h, const double* X, int64_t N, int D, const double* ls, double sf2, const double* alpha,
                         int P, const double* Kinv, int64_t ldk, double* dout) {
  GPK_REQUIRE(h, X && ls && alpha && Kinv, "lml_grad: null pointer");
  GPK_REQUIRE(h, N >= 1 && D >= 1 && D <= DMAXG && P >= 1 && P <= GPK_MAX_P, "lml_grad: D must be in [1, 16], P in [1, 16]");
  const int64_t Np = gpk_padded(N);
  GPK_REQUIRE(h, ldk >= Np && ldk % 2 == 0 && ((uintptr_t)Kinv % 16) == 0, "lml_grad: Kinv must be padded and 16-byte aligned");
  LsG l;
  for (int d = 0; d < DMAXG; ++d) l.v[d] = 1.0;
  for (int d = 0; d < D; ++d) {
    GPK_REQUIRE(h, ls[d] > 0.0, "lml_grad: length-scales must be positive");
    l.v[d] = ls[d];
  }
  const int64_t nt = Np / TS, ntiles = nt * (nt + 1) / 2;
  const int nblocks = (int)(ntiles < 4096 ? ntiles : 4096);
  void* ws = nullptr;
  GPK_TRY(gpk_scratch(h, (size_t)(nblocks + 1) * GW * sizeof(double), &ws));
  double* partial = (double*)ws;
  gpk_time_begin(h, GPK_TIMED_GRAD);
  hipLaunchKernelGGL(lml_grad_kernel, dim3(nblocks), dim3(256), 0, h->stream, X, (long long)N, D, l, sf2, alpha, P,
                     Kinv, (long long)ldk, (long long)ntiles, partial);
  gpk_time_end(h);
  GPK_LAUNCH_CHECK(h);
  hipLaunchKernelGGL(grad_reduce_kernel, dim3(GW), dim3(256), 0, h->stream, (const double*)partial, nblocks, dout);
  GPK_LAUNCH_CHECK(h);
  return GPK_OK;
}

static void grad_from_sums(const double* host, int D, double noise, double* grad) {
  for (int d = 0; d < D; ++d) grad[d] = 0.5 * host[d];
  grad[D] = 0.5 * noise * host[DMAXG];
  grad[D + 1] = 0.5 * host[DMAXG + 1];
}

extern "C" int gpk_lml_grad(gpk_handle h, const double* X, int64_t N, int D, const double* ls, double sf2,
                            double noise, const double* alpha, int P, const double* Kinv, int64_t ldk,
                            double* grad) {
  if (!h) return GPK_BAD_ARG;
  GPK_REQUIRE(h, grad, "lml_grad: null pointer");
  double* dout = h->d_small + 64;
  GPK_TRY(gpk_lml_grad_enqueue(h, X, N, D, ls, sf2, alpha, P, Kinv, ldk, dout));
  GPK_CHECK_HIP(h, hipStreamSynchronize(h->stream));            // (d_small is the pinned block h_small itself)
  grad_from_sums(h->h_small + 64, D, noise, grad);
  return GPK_OK;
}

int gpk_lml_chain_batched(gpk_handle h, int B, const double* X, int64_t N, int D, const double* ls, const double* sf2,
                          const double* noise, const double* Yn, int64_t Ne, double* K, int64_t Np, double* winv, double* W,
                          double* T, size_t tsz, double* alpha, double* Kinv, double* terms, double* grads, int* info) {
  GPK_REQUIRE(h, B >= 1 && B <= GPK_MAX_BATCH && h->batch == 1, "lml_batched: bad batch");
  const size_t nn = (size_t)Np * Np;
  GPK_TRY(gpk_batch_begin(h, B));
  auto chain = [&]() -> int {
    const struct { const void* p; size_t stride; } bufs[] = {
        {K, nn * 8}, {winv, (size_t)Np * GPK_TILE * 8}, {W, nn * 8}, {T, tsz * 8}, {Yn, (size_t)Ne * 8}, {alpha, (size_t)Ne * 8},
        {Kinv, nn * 8}};
    for (const auto& b : bufs)
      if (b.p) GPK_TRY(gpk_batch_buffer(h, b.p, (int64_t)b.stride));
    // factor and inverse factor: one persistent launch for all problems when a gradient is wanted and the size allows
    int fused = 0;
    if (Kinv) GPK_TRY(gpk_potrf_trtri_enqueue(h, K, Np, Np, winv, W, Np, Kinv, &fused));
    if (!fused) {
      GPK_TRY(gpk_potrf_enqueue(h, K, Np, Np, winv));
      GPK_TRY(gpk_trtri(h, K, Np, Np, winv, W, Np, T));
    }
    GPK_TRY(gpk_potrs_inv(h, W, Np, Np, Yn, N, 1, alpha));
    if (Kinv) GPK_TRY(gpk_wtw(h, W, Np, Np, Kinv, Np));
    // per problem: terms to d_small[4 b ..], gradient sums to d_small[64 + 32 b ..]; then everything back in one copy
    for (int b = 0; b < B; ++b) {
      GPK_TRY(gpk_lml_terms_enqueue(h, K + b * nn, N, Np, Yn + (size_t)b * Ne, alpha + (size_t)b * Ne, 1, h->d_small + 4 * b,
                                    b == B - 1));          // (the last one also writes the chain's status words)
      if (Kinv)
        GPK_TRY(gpk_lml_grad_enqueue(h, X, N, D, ls + (size_t)b * D, sf2[b], alpha + (size_t)b * Ne, 1, Kinv + b * nn, Np,
                                     h->d_small + 64 + 32 * b));
    }
    // (terms, gradient sums and status words are written straight into the pinned block: one synchronisation, no copy)
    GPK_CHECK_HIP(h, hipStreamSynchronize(h->stream));
    const int* hst = reinterpret_cast<const int*>(h->h_small + GPK_STATUS_OFF);
    const int rc = gpk_potrf_finish(h, hst, info, hst[GPK_MAX_BATCH]);
    if (rc != GPK_OK && rc != GPK_NOT_PD) return rc;          // per-problem outcome is in info[]
    for (int b = 0; b < B; ++b) {
      terms[2 * b] = h->h_small[4 * b];
      terms[2 * b + 1] = h->h_small[4 * b + 1];
      if (Kinv) grad_from_sums(h->h_small + 64 + 32 * b, D, noise[b], grads + (size_t)b * (D + 2));
    }
    return GPK_OK;
  };
  const int rc = chain();
  (void)gpk_batch_end(h);
  return rc;
}

// One evaluation of the log-marginal likelihood (and its gradient) as ONE chain of launches with ONE synchronisation: what the
// estimator's optimiser loop runs per trial theta.  At the reference's own size (N = 1000) the three host round trips of the
// call-by-call route - the pivot check of gpk_potrf, gpk_lml_terms, gpk_lml_grad - are a seventh of an evaluation.
extern "C" int gpk_lml_eval(gpk_handle h, const double* X, int64_t N, int D, const double* ls, double sf2, double diag_add,
                            double noise, const double* Yn, int P, double* K, int64_t Np, double* winv, double* W, double* work,
                            double* alpha, double* Kinv, double* terms, double* grad, int* info) {
  if (!h) return GPK_BAD_ARG;
  GPK_REQUIRE(h, X && ls && Yn && K && winv && W && work && alpha && terms && info, "lml_eval: null pointer");
  GPK_REQUIRE(h, h->batch == 1, "lml_eval: not available in batched mode");
  GPK_REQUIRE(h, Np == gpk_padded(N) && P >= 1 && P <= GPK_MAX_P, "lml_eval: bad sizes");
  GPK_REQUIRE(h, grad == nullptr || Kinv != nullptr, "lml_eval: the gradient needs the Kinv buffer");
  GPK_TRY(gpk_gram(h, GPK_F64, X, N, D, ls, sf2, diag_add, K, Np));
  // small matrices (Kinv given: it is free until W^T W writes it - a value-only evaluation may pass it as scratch just for this):
  // factor and inverse factor as ONE persistent launch
  int fused = 0;
  if (Kinv) GPK_TRY(gpk_potrf_trtri_enqueue(h, K, Np, Np, winv, W, Np, Kinv, &fused));
  if (!fused) {
    GPK_TRY(gpk_potrf_enqueue(h, K, Np, Np, winv));
    GPK_TRY(gpk_trtri(h, K, Np, Np, winv, W, Np, work));
  }
  GPK_TRY(gpk_potrs_inv(h, W, Np, Np, Yn, N, P, alpha));
  double* dterms = h->d_small;            // [0, 1 + P): the terms; [64, 64 + GW): the gradient sums; then the pivot failure
  double* dgrad = h->d_small + 64;
  GPK_TRY(gpk_lml_terms_enqueue(h, K, N, Np, Yn, alpha, P, dterms, 1));    // (+ the status words: pivot failure, "gave up")
  if (grad) {
    GPK_TRY(gpk_wtw(h, W, Np, Np, Kinv, Np));
    GPK_TRY(gpk_lml_grad_enqueue(h, X, N, D, ls, sf2, alpha, P, Kinv, Np, dgrad));
  }
  // terms, gradient sums, the pivot failure and the one-launch factorisation's flag were written straight into the pinned,
  // device-mapped block: ONE synchronisation, no copy command, no status launch
  GPK_CHECK_HIP(h, hipStreamSynchronize(h->stream));
  const int* hst = reinterpret_cast<const int*>(h->h_small + GPK_STATUS_OFF);
  GPK_TRY(gpk_potrf_finish(h, hst, info, hst[GPK_MAX_BATCH]));   // GPK_NOT_PD: what follows the factorisation ran on a finite, meaningless factor
  for (int i = 0; i < 1 + P; ++i) terms[i] = h->h_small[i];
  if (grad) grad_from_sums(h->h_small + 64, D, noise, grad);
  return GPK_OK;
}


// ---- kernel matrix and its length-scale derivative for a pair of point sets (the package kernel object's gradient) --------
// K[i][j] = sf2 exp(-r2_ij / 2), Q[i][j] = K[i][j] r2_ij with r2_ij = sum_d ((x1_id - x2_jd) / ls_d)^2 (exact differences).
// For the isotropic kernel of gaussian_process.py:43-60: dK/dl = K d2 / l^3 = Q / l and dK/dsf2 = K / sf2.
// One thread per entry, 16 x 16 entries per workgroup, the two point blocks staged in LDS; an inspection call, not a hot path.
namespace {
__global__ __launch_bounds__(256) void rbf_kernel_grad_kernel(const double* __restrict__ X1, long long n1,
                                                              const double* __restrict__ X2, long long n2, int D, LsG ls,
                                                              double sf2, double* __restrict__ K, double* __restrict__ Q,
                                                              long long ld) {
  __shared__ double a[16][DMAXG + 1], b[16][DMAXG + 1];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const long long i0 = (long long)blockIdx.y * 16, j0 = (long long)blockIdx.x * 16;
  if (tx < D) {
    a[ty][tx] = i0 + ty < n1 ? X1[(i0 + ty) * D + tx] / ls.v[tx] : 0.0;
    b[ty][tx] = j0 + ty < n2 ? X2[(j0 + ty) * D + tx] / ls.v[tx] : 0.0;
  }
  __syncthreads();
  const long long i = i0 + ty, j = j0 + tx;
  if (i >= n1 || j >= n2) return;
  double r2 = 0.0;
  for (int d = 0; d < D; ++d) {
    const double t = a[ty][d] - b[tx][d];
    r2 = __builtin_fma(t, t, r2);
  }
  const double k = sf2 * exp_neg64(-0.5 * r2);
  K[i * ld + j] = k;
  if (Q) Q[i * ld + j] = k * r2;
}
}  // namespace

extern "C" int gpk_rbf_kernel_grad(gpk_handle h, const double* X1, int64_t n1, const double* X2, int64_t n2, int D,
                                   const double* ls, double sf2, double* K, double* Q, int64_t ld) {
  if (!h) return GPK_BAD_ARG;
  GPK_REQUIRE(h, X1 && X2 && K && ls, "rbf_kernel_grad: null pointer");
  GPK_REQUIRE(h, n1 >= 1 && n2 >= 1 && ld >= n2, "rbf_kernel_grad: empty input or ld < n2");
  GPK_REQUIRE(h, D >= 1 && D <= DMAXG, "rbf_kernel_grad: D must be in [1, 16]");
  GPK_REQUIRE(h, (n1 + 15) / 16 < 65536, "rbf_kernel_grad: n1 too large");
  LsG l{};
  for (int d = 0; d < DMAXG; ++d) l.v[d] = d < D ? ls[d] : 1.0;
  for (int d = 0; d < D; ++d) GPK_REQUIRE(h, ls[d] > 0.0, "rbf_kernel_grad: length scales must be positive");
  hipLaunchKernelGGL(rbf_kernel_grad_kernel, dim3((unsigned)((n2 + 15) / 16), (unsigned)((n1 + 15) / 16)), dim3(256), 0,
                     h->stream, X1, (long long)n1, X2, (long long)n2, D, l, sf2, K, Q, (long long)ld);
  GPK_LAUNCH_CHECK(h);
  return GPK_OK;
}
