// Small-batch serving kernels (fp64; M <= 32 queries, D <= 16, P <= 16, Np <= 16384): the control loop's
// single-row `predict_residual` and horizon-25 calls (simple_gp.py:175-206, gaussian_process.py:199-240).
// The general chain spends seven launches (copy, fused mean, mean reduce, K*^T, tile GEMM on a 128-query
// panel, column-sum reduce, finalize = 80 us of GPU time for 25 queries at N = 1000); here it is two:
//
//   small_cross_mean_kernel  32 training rows per workgroup: K*[m][j] = sf2 exp(-|x_j - q_m|^2 / 2) (exact
//                            differences of inputs divided by the length-scale, as everywhere else), written
//                            query-major for the second kernel, and the workgroup's share of K* alpha; the last
//                            workgroup to finish adds the shares in a fixed order and writes the means.
//   small_var_kernel         16 rows of W = L^-1 per workgroup: V = W K*^T as a 16 x (16|32) x k product on
//                            v_mfma_f64_16x16x4_f64 (both operands read straight from L2: each lane's
//                            16-byte pieces are k-contiguous), the k-range split over the 4 waves; squares
//                            summed over the rows; the last workgroup adds the shares and writes
//                            max(kss - sum, floor).
//
// Both results land in the caller's (pinned, mapped) output block; the queries are read from it as well.
#include "gpk_internal.h"
#include "gpk_math.h"

namespace {

constexpr int SQ = GPK_SMALL_MAX_M, SJ = 32, SD = 16, SP = 16, SR = 16;
struct Arr16 { double v[16]; };
// Per-model parameters (model = blockIdx.y).  One model with P <= 16 outputs, or B <= 8 single-output models that
// share the query batch (the per-axis GPs of gp_trainer.py): output o = model * P + p indexes ymean / ystd.
struct SmallK {
  const double* X[GPK_SMALL_MAX_MODELS];
  const double* alpha[GPK_SMALL_MAX_MODELS];
  const double* W[GPK_SMALL_MAX_MODELS];
  double ls[GPK_SMALL_MAX_MODELS][16];
  double sf2[GPK_SMALL_MAX_MODELS], kss[GPK_SMALL_MAX_MODELS];
  double ymean[16], ystd[16];
};
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2v __attribute__((ext_vector_type(2)));

// Every workgroup has stored its share; true (in all its threads) for the last one to get here.  The barrier
// orders the workgroup's stores before thread 0's ticket; the ticket is one acquire-release atomic at device scope
// (release: the shares are written back before it; acquire: the last workgroup drops its cached lines before it
// reads the others' shares) -- one cache write-back per workgroup instead of one per wave.  The counter is left at
// zero for the next launch.
__device__ __forceinline__ bool last_workgroup(unsigned* counter, int tid) {
  __shared__ int is_last;
  __syncthreads();
  if (tid == 0) {
    const unsigned t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    is_last = (t == gridDim.x - 1);
    if (is_last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  return is_last != 0;
}

// Sum of `count` shares p[(first + k * step) * stride], k = 0.., in that order; the loads go out eight at a time
// (the last workgroup reads them from memory: one dependent load per share would cost a miss latency each).
__device__ __forceinline__ double sum_shares(const double* p, unsigned first, unsigned step, unsigned shares,
                                             long long stride) {
  double s = 0.0;
  for (unsigned g = first; g < shares; g += 8 * step) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const unsigned gu = g + u * step;
      v[u] = p[(long long)min(gu, shares - 1) * stride];
      if (gu >= shares) v[u] = 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  return s;
}

// The fixed-order sum of the mean shares, by whichever workgroup finishes last.  `lds`: 4 * SQ * SP doubles.  The
// grouping of the sum depends on M * P only (256 threads take part whatever the workgroup size), so a mean-only
// call and a mean + variance call return the same bits.
__device__ __forceinline__ void finish_means(const double* pmean, unsigned shares, int M, int P, const double* ymean,
                                             const double* ystd, double* mean_out, int tid, double* lds) {
  constexpr int NT = 256;
  const int MP = M * P;
  const int nparts = 4 * MP <= NT ? 4 : (2 * MP <= NT ? 2 : 1);
  const int chunk = NT / nparts, part = tid / chunk, tl = tid - part * chunk;
  for (int base = 0; base < MP; base += chunk) {
    const int t = base + tl;
    if (tid < NT && t < MP) lds[part * (SQ * SP) + t] = sum_shares(pmean + t, part, nparts, shares, SQ * SP);
    __syncthreads();
    if (part == 0 && t < MP) {
      double s = lds[t];
      for (int k = 1; k < nparts; ++k) s += lds[k * (SQ * SP) + t];
      const int p = t % P;
      mean_out[t] = ymean[p] + ystd[p] * s;
    }
    if (base + chunk < MP) __syncthreads();      // another pass reuses the scratch
  }
}

// FINISH: this launch is the only one (mean-only call) and elects the workgroup that writes the means; otherwise
// small_var_kernel's last workgroup does it.
template <bool FINISH>
__global__ __launch_bounds__(256) void small_cross_mean_kernel(SmallK k, long long N, long long Np, int D, int P,
                                                               const double* __restrict__ Xq, int M, double* Ks,
                                                               double* pmean, unsigned* counter, double* mean_out) {
  __shared__ double q[SQ][SD + 1];
  __shared__ double ks[SQ][SJ + 1];
  __shared__ double al[SJ][SP + 1];
  const int b = blockIdx.y;
  const double* __restrict__ X = k.X[b];
  const double* __restrict__ alpha = k.alpha[b];
  const double* ls = k.ls[b];
  const double sf2 = k.sf2[b];
  if (Ks) Ks += (long long)b * SQ * Np;
  pmean += (long long)b * gridDim.x * (SQ * SP);
  mean_out += (long long)b * M * P;
  counter += b;
  const int tid = threadIdx.x, jl = tid & 31, mg = tid >> 5;
  const long long j0 = (long long)blockIdx.x * SJ, j = j0 + jl;
  const bool valid = j < N;
  // all global reads first (the queries may sit in host memory: the longest latency), then their uses
  double qv[2], av[2], xr[SD];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int e = tid + 256 * u;
    qv[u] = (e < M * D) ? Xq[e] : 0.0;
    const int jj = e / P;
    av[u] = (e < SJ * P && j0 + jj < N) ? alpha[j0 * P + e] : 0.0;
  }
#pragma unroll
  for (int d = 0; d < SD; ++d) xr[d] = (d < D && valid) ? X[j * D + d] : 0.0;
#pragma unroll
  for (int d = 0; d < SD; ++d) xr[d] = xr[d] / ls[d];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int e = tid + 256 * u;
    if (e < SJ * P) al[e / P][e % P] = av[u];
    if (e < M * D) q[e / D][e % D] = qv[u] / ls[e % D];
  }
  __syncthreads();
#pragma unroll
  for (int mi = 0; mi < SQ / 8; ++mi) {
    const int m = mg + 8 * mi;
    if (m < M) {
      double d2 = 0.0;
#pragma unroll
      for (int d = 0; d < SD; ++d)
        if (d < D) {
          const double df = q[m][d] - xr[d];
          d2 = __builtin_fma(df, df, d2);
        }
      const double v = valid ? sf2 * gpk_exp_neg(-0.5 * d2) : 0.0;
      ks[m][jl] = v;
      if (Ks) Ks[(long long)m * Np + j] = v;
    }
  }
  __syncthreads();
  for (int t = tid; t < M * P; t += 256) {
    const int m = t / P, p = t - m * P;
    double s = 0.0;
#pragma unroll 8
    for (int jj = 0; jj < SJ; ++jj) s = __builtin_fma(ks[m][jj], al[jj][p], s);
    pmean[(long long)blockIdx.x * (SQ * SP) + t] = s;
  }
  if constexpr (FINISH) {
    if (last_workgroup(counter, tid)) {
      __shared__ double fin[4 * SQ * SP];
      finish_means(pmean, gridDim.x, M, P, k.ymean + b * P, k.ystd + b * P, mean_out, tid, fin);
    }
  }
}

// NMB = 1: up to 16 queries, 2: up to 32.  8 waves: wave w takes the 64-wide k-chunks w, w + 8, ..., the loads of
// the next one in flight while the current one is multiplied.
constexpr int VW = 8;

template <int NMB>
struct VFrag { d2v a[8], b[NMB][8]; };

template <int NMB>
__device__ __forceinline__ void vload(VFrag<NMB>& f, const double* wp, const double* const (&kp)[NMB], long long kc) {
#pragma unroll
  for (int s = 0; s < 8; ++s) f.a[s] = *reinterpret_cast<const d2v*>(wp + kc + 8 * s);
#pragma unroll
  for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
    for (int s = 0; s < 8; ++s) f.b[mb][s] = *reinterpret_cast<const d2v*>(kp[mb] + kc + 8 * s);
}

template <int NMB>
__device__ __forceinline__ void vmul(const VFrag<NMB>& f, long long kc, int kq, long long row, d4 (&acc)[NMB]) {
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const long long k0 = kc + 8 * s + 2 * kq;
    const double ax = (k0 <= row) ? f.a[s].x : 0.0;            // strictly-upper entries never enter
    const double ay = (k0 + 1 <= row) ? f.a[s].y : 0.0;
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) {
      acc[mb] = __builtin_amdgcn_mfma_f64_16x16x4f64(ax, f.b[mb][s].x, acc[mb], 0, 0, 0);
      acc[mb] = __builtin_amdgcn_mfma_f64_16x16x4f64(ay, f.b[mb][s].y, acc[mb], 0, 0, 0);
    }
  }
}

template <int NMB>
__global__ __launch_bounds__(64 * VW) void small_var_kernel(SmallK k, long long ldw, long long Np, const double* Ks,
                                                            int M, int P, double floor_, const double* pmean,
                                                            unsigned mean_shares, double* pvar, unsigned* counter,
                                                            double* mean_out, double* var_out) {
  __shared__ double red[VW][NMB][16][17];
  __shared__ double sq[NMB][16][17];
  const int b = blockIdx.y;
  const double* __restrict__ W = k.W[b];
  const double kss = k.kss[b];
  Ks += (long long)b * SQ * Np;
  pmean += (long long)b * mean_shares * (SQ * SP);
  pvar += (long long)b * gridDim.x * SQ;
  mean_out += (long long)b * M * P;
  var_out += (long long)b * M;
  counter += b;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i = lane & 15, kq = lane >> 4;
  const long long r0 = (long long)blockIdx.x * SR, row = r0 + i;
  const long long kend = min(Np, (r0 + SR + 63) / 64 * 64);   // the rows' diagonal, rounded up to the chunk
  const int nch = (int)(kend / 64);
  const double* wp = W + row * ldw + 2 * kq;
  const double* kp[NMB];
#pragma unroll
  for (int mb = 0; mb < NMB; ++mb) kp[mb] = Ks + (long long)min(16 * mb + i, M - 1) * Np + 2 * kq;  // columns >= M: unused
  d4 acc[NMB];
#pragma unroll
  for (int mb = 0; mb < NMB; ++mb) acc[mb] = d4{0.0, 0.0, 0.0, 0.0};
  // Chunk of 64 k: step s covers k = 8 s + 2 kq + {0, 1} for kq = 0..3 -- one 16-byte load per lane and
  // operand, 64 contiguous bytes per row and step; the two halves feed two MFMAs.
  VFrag<NMB> f0, f1;
  if (w < nch) vload<NMB>(f0, wp, kp, (long long)w * 64);
  for (int c = w; c < nch; c += 2 * VW) {
    const bool n1 = c + VW < nch, n2 = c + 2 * VW < nch;
    if (n1) vload<NMB>(f1, wp, kp, (long long)(c + VW) * 64);
    vmul<NMB>(f0, (long long)c * 64, kq, row, acc);
    if (n2) vload<NMB>(f0, wp, kp, (long long)(c + 2 * VW) * 64);
    if (n1) vmul<NMB>(f1, (long long)(c + VW) * 64, kq, row, acc);
  }
  // accumulator map: column = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
  for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[w][mb][kq + 4 * r][i] = acc[mb][r];
  __syncthreads();
  for (int e = tid; e < NMB * 256; e += 64 * VW) {
    const int mb = e >> 8, r = (e >> 4) & 15, c = e & 15;
    double v = 0.0;
#pragma unroll
    for (int u = 0; u < VW; ++u) v += red[u][mb][r][c];
    sq[mb][r][c] = v * v;
  }
  __syncthreads();
  if (tid < NMB * 16) {
    const int mb = tid >> 4, c = tid & 15;
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += sq[mb][r][c];
    pvar[(long long)blockIdx.x * SQ + tid] = s;
  }
  // The mean shares are complete since the previous launch: the workgroup with the shortest rows adds them up while
  // the others are still multiplying, off the critical path (scratch: the reduction buffer, free by now).
  static_assert(VW * 16 * 17 >= 4 * SQ * SP, "the reduction buffer doubles as the mean scratch");
  if (blockIdx.x == 0) finish_means(pmean, mean_shares, M, P, k.ymean + b * P, k.ystd + b * P, mean_out, tid, &red[0][0][0][0]);
  if (last_workgroup(counter, tid)) {
    __shared__ double part[2 * VW][SQ];
    const int m = tid & 31, pt = tid >> 5;
    part[pt][m] = (m < NMB * 16) ? sum_shares(pvar + m, pt, 2 * VW, gridDim.x, SQ) : 0.0;
    __syncthreads();
    if (tid < M) {
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < 2 * VW; ++k) t += part[k][tid];
      var_out[tid] = fmax(kss - t, floor_);
    }
  }
}

}  // namespace

size_t gpk_small_work_doubles(int64_t Np, int B) { return (size_t)B * Np * (SQ + SQ * SP / SJ + SQ / SR); }

bool gpk_small_ok(int64_t Np, int D, int P, int64_t M) {
  return M >= 1 && M <= SQ && D >= 1 && D <= SD && P >= 1 && P <= SP && Np <= GPK_SMALL_MAX_NP;
}

// B models (blockIdx.y) x P outputs each (B > 1 requires P == 1).  X / alpha / W: B device pointers; ls: B x D;
// sf2, kss: B; y_mean, y_std: B * P.  mean_out: (B, M, P), var_out: (B, M) or null.
int gpk_small_predict(gpk_handle h, int B, const double* const* X, const double* const* alpha, int64_t N, int D, int P,
                      const double* ls, const double* sf2, const double* y_mean, const double* y_std,
                      const double* const* W, int64_t Np, int64_t ldw, const double* kss, double floor_,
                      const double* Xq, int64_t M, double* work, double* mean_out, double* var_out) {
  GPK_REQUIRE(h, B >= 1 && B <= GPK_SMALL_MAX_MODELS && (B == 1 || P == 1), "small predict: 1 model, or up to 8 single-output models");
  GPK_REQUIRE(h, gpk_small_ok(Np, D, P, M) && Np == gpk_padded(N), "small predict: shape outside the small-batch path");
  GPK_REQUIRE(h, !var_out || (W && ldw >= Np && ldw % 2 == 0), "small predict: variance needs the inverse factor");
  SmallK k{};
  for (int b = 0; b < B; ++b) {
    GPK_REQUIRE(h, X[b] && alpha[b] && (!var_out || W[b]), "small predict: null model pointer");
    GPK_REQUIRE(h, !var_out || ((uintptr_t)W[b] % 16) == 0, "small predict: the inverse factor must be 16-byte aligned");
    k.X[b] = X[b]; k.alpha[b] = alpha[b]; k.W[b] = var_out ? W[b] : nullptr;
    for (int d = 0; d < 16; ++d) k.ls[b][d] = 1.0;
    for (int d = 0; d < D; ++d) {
      GPK_REQUIRE(h, ls[b * D + d] > 0.0, "length-scales must be positive");
      k.ls[b][d] = ls[b * D + d];
    }
    k.sf2[b] = sf2[b];
    k.kss[b] = var_out ? kss[b] : 0.0;
  }
  for (int o = 0; o < B * P; ++o) { k.ymean[o] = y_mean[o]; k.ystd[o] = y_std[o]; }
  const unsigned ga = (unsigned)(Np / SJ), gb = (unsigned)(Np / SR);
  double* Ks = work;                                       // B x SQ x Np
  double* pmean = Ks + (size_t)B * SQ * Np;                // B x ga x (SQ * SP)
  double* pvar = pmean + (size_t)B * ga * (SQ * SP);       // B x gb x SQ
  if (var_out) {
    hipLaunchKernelGGL(small_cross_mean_kernel<false>, dim3(ga, B), dim3(256), 0, h->stream, k, (long long)N, (long long)Np,
                       D, P, Xq, (int)M, Ks, pmean, h->d_count, mean_out);
    GPK_LAUNCH_CHECK(h);
    if (M <= 16)
      hipLaunchKernelGGL(small_var_kernel<1>, dim3(gb, B), dim3(64 * VW), 0, h->stream, k, (long long)ldw, (long long)Np, Ks,
                         (int)M, P, floor_, pmean, ga, pvar, h->d_count + GPK_SMALL_MAX_MODELS, mean_out, var_out);
    else
      hipLaunchKernelGGL(small_var_kernel<2>, dim3(gb, B), dim3(64 * VW), 0, h->stream, k, (long long)ldw, (long long)Np, Ks,
                         (int)M, P, floor_, pmean, ga, pvar, h->d_count + GPK_SMALL_MAX_MODELS, mean_out, var_out);
  } else {
    hipLaunchKernelGGL(small_cross_mean_kernel<true>, dim3(ga, B), dim3(256), 0, h->stream, k, (long long)N, (long long)Np,
                       D, P, Xq, (int)M, (double*)nullptr, pmean, h->d_count, mean_out);
  }
  GPK_LAUNCH_CHECK(h);
  return GPK_OK;
}
