// Blocked Cholesky, triangular solves and K^-1 on top of the MFMA tile GEMM.
//
// Everything works on matrices padded to whole 128 x 128 tiles (identity in the padding).
// The factorisation is recursive: potrf(n) = potrf(n/2), trsm, syrk, potrf(n - n/2), so
// almost all flops land in large fp64-MFMA GEMM launches; the recursion bottoms out in a
// 128 x 128 leaf factorised inside one workgroup's LDS.  Each leaf's inverse is kept
// (winv, Np x 128) so that every triangular solve is a chain of GEMMs.
#include <vector>

#include "gpk_internal.h"

namespace {

constexpr int NB = 128;       // leaf size

// ---- leaf: Cholesky factor AND inverse of one 128 x 128 diagonal block, one workgroup ------------
// The block lives in LDS (row stride 130 doubles: conflict-free ds_read_b64 for the MFMA operand
// pattern row = lane & 15, k = lane >> 4).  Blocked left-looking sweep with 16 x 16 sub-blocks:
//   P1  block column jb  -= L[ib, 0:jb] L[jb, 0:jb]^T          fp64 MFMA 16x16x4, one block per wave
//   P2  16 x 16 diagonal block AND its inverse by one wave: a row per lane (16 block rows + 16 identity rows) in
//       registers, right-looking sweep with the pivot column broadcast by v_mov_b64_dpp row_newbcast (no LDS and no
//       scalar round trips in the loop)
//   P3  rows below: X = A W_jj^T on the MFMA (the solve x L_jj^T = a through the explicit 16 x 16 inverse)
// then the inverse W = L^-1 block diagonal by block diagonal on the MFMA:
//   W_ij = -W_ii (sum_{k=j}^{i-1} L_ik W_kj),   i - j = 1 .. 7
// The inner sum's accumulator is used directly as the B operand of the second product (the f64
// C/D map row = (lane >> 4) + 4 reg is exactly the B-operand map of k-step reg).  Off-diagonal W
// blocks are parked transposed in the (otherwise unused) upper triangle of the LDS image.
// FACTOR == false: the block already holds a factor (imported model); only W is produced.
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int LSA = 130;   // LDS row stride of the 128 x 128 image
constexpr int LSW = 18;    // row stride of the 16 x 16 diagonal inverses

// value of `x` in lane SRC of the caller's 16-lane DPP row, in every lane of that row: one v_mov_b64_dpp
// (the DPP control must be a literal for the builtin to keep its 64-bit type inside a template: one case per lane)
template <int SRC>
__device__ __forceinline__ double row_bcast(double x) {
  const long long b = __builtin_bit_cast(long long, x);
  long long r = 0;
#define GPK_ROW_NEWBCAST(n) if constexpr (SRC == n) r = __builtin_amdgcn_update_dpp(0ll, b, 0x150 + n, 0xf, 0xf, true);
  GPK_ROW_NEWBCAST(0) GPK_ROW_NEWBCAST(1) GPK_ROW_NEWBCAST(2) GPK_ROW_NEWBCAST(3) GPK_ROW_NEWBCAST(4) GPK_ROW_NEWBCAST(5)
  GPK_ROW_NEWBCAST(6) GPK_ROW_NEWBCAST(7) GPK_ROW_NEWBCAST(8) GPK_ROW_NEWBCAST(9) GPK_ROW_NEWBCAST(10) GPK_ROW_NEWBCAST(11)
  GPK_ROW_NEWBCAST(12) GPK_ROW_NEWBCAST(13) GPK_ROW_NEWBCAST(14) GPK_ROW_NEWBCAST(15)
#undef GPK_ROW_NEWBCAST
  static_assert(SRC >= 0 && SRC < 16, "row_newbcast lane");
  return __builtin_bit_cast(double, r);
}
template <int V> struct LeafC { static constexpr int value = V; };
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(LeafC<I>{});
    static_for<I + 1, N>(f);
  }
}

template <bool FACTOR>
__global__ __launch_bounds__(256) void leaf_kernel(double* __restrict__ A0, long long lda, int row0,
                                                   int* __restrict__ info0, double* __restrict__ W0,
                                                   long long strideA, long long strideW) {
  // one workgroup per problem of the batch (strides in bytes)
  double* __restrict__ A = reinterpret_cast<double*>(reinterpret_cast<char*>(A0) + blockIdx.x * strideA);
  double* __restrict__ W = reinterpret_cast<double*>(reinterpret_cast<char*>(W0) + blockIdx.x * strideW);
  int* __restrict__ info = info0 + blockIdx.x;
  __shared__ __attribute__((aligned(16))) double a[NB * LSA];
  __shared__ __attribute__((aligned(16))) double wd[8 * 16 * LSW];   // wd[b][r][c] = W_bb[c][r]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  // The block comes in as 16-byte pieces, all 32 loads of a thread in flight at once (the kernel is one workgroup on
  // one CU: its cost is latency, and 64 dependent 8-byte round trips were a third of it).  The strictly upper part is
  // read too (valid memory: K is symmetric there) and zeroed on the way into LDS.
  typedef double dv2 __attribute__((ext_vector_type(2)));
  {
    dv2 buf[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      const int e = tid + 256 * u, i = e >> 6, j = (e & 63) * 2;
      buf[u] = *reinterpret_cast<const dv2*>(A + (long long)i * lda + j);
    }
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      const int e = tid + 256 * u, i = e >> 6, j = (e & 63) * 2;
      dv2 v = buf[u];
      if (j > i) v.x = 0.0;
      if (j + 1 > i) v.y = 0.0;
      *reinterpret_cast<dv2*>(a + i * LSA + j) = v;
    }
  }
  __syncthreads();

  for (int jb = 0; jb < 8; ++jb) {
    const int c0 = 16 * jb;
    if (FACTOR) {
      // ---- P1: left-looking update of block column jb
      if (jb > 0) {
        for (int ib = jb + wave; ib < 8; ib += 4) {
          d4 acc;
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[r] = a[(16 * ib + lq + 4 * r) * LSA + c0 + lr];
          for (int kb = 0; kb < jb; ++kb) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              const double av = a[(16 * ib + lr) * LSA + 16 * kb + 4 * s + lq];
              const double bv = -a[(c0 + lr) * LSA + 16 * kb + 4 * s + lq];
              acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
            }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) a[(16 * ib + lq + 4 * r) * LSA + c0 + lr] = acc[r];
        }
        __syncthreads();
      }
    }
    // ---- P2: the 16 x 16 diagonal block AND its inverse by ONE wave, in registers, with no LDS and no scalar
    // round trips inside the loop.  Every 16-lane DPP row holds a copy of the block, lane (i) = row i in 16 registers
    // (u); row 1 (lanes 16..31) also carries the identity rows e_i (x).  Right-looking sweep with unscaled columns,
    // a_ik -= (a_ic / d_c) a_kc: what a step needs from another row -- the pivot d_c and the column entries a_kc --
    // is one v_mov_b64_dpp row_newbcast each (lane k of the own 16-lane row to all its lanes).  The identity rows
    // undergo the same updates (x L_jj^T = e_r by substitution), which yields W_jj^T = L_jj^-T.  One final scaling
    // of the columns by 1 / sqrt(d_c) turns both into L_jj and W_jj^T (d_c / sqrt(d_c) = sqrt(d_c) on the diagonal).
    // Straight-line and branch-free (a failed pivot is handled by selects and reported once); the next pivot's
    // reciprocal -- v_rcp_f64 + two Newton steps, error ~1e-16, the loop-carried chain -- is issued right after the
    // one update it depends on.  Entries above the diagonal of the block rows are never read.
    if (wave == 0) {
      const int i = lane & 15;
      double u[16], x[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const double l = a[(c0 + i) * LSA + c0 + c];
        u[c] = (c <= i) ? l : 0.0;
        x[c] = (c == i) ? 1.0 : 0.0;
      }
      if (FACTOR) {
        int bad = 0;                             // 1-based column of the first non-positive pivot
        double d = row_bcast<0>(u[0]), rd;
        auto pivot = [&](auto cc) {              // d holds the (final) pivot of column C
          constexpr int C = decltype(cc)::value;
          const bool ok = d > 0.0;
          bad = (!ok && bad == 0) ? C + 1 : bad;
          u[C] = (!ok && i == C) ? 1.0 : u[C];
          d = ok ? d : 1.0;
          rd = __builtin_amdgcn_rcp(d);
          rd = __builtin_fma(__builtin_fma(-d, rd, 1.0), rd, rd);
          rd = __builtin_fma(__builtin_fma(-d, rd, 1.0), rd, rd);
        };
        pivot(LeafC<0>{});
        static_for<0, 16>([&](auto cc) {
          constexpr int C = decltype(cc)::value;
          const double nf = -u[C] * rd, ng = -x[C] * rd;
          if constexpr (C + 1 < 16) {
            const double s1 = row_bcast<C + 1>(u[C]);
            u[C + 1] = __builtin_fma(nf, s1, u[C + 1]);
            x[C + 1] = __builtin_fma(ng, s1, x[C + 1]);
            d = row_bcast<C + 1>(u[C + 1]);      // the next pivot is final now
            pivot(LeafC<C + 1>{});
          }
          double sk[16];                          // column C across the row, read ahead of the updates that use it
          static_for<C + 2, 16>([&](auto kk) { sk[decltype(kk)::value] = row_bcast<decltype(kk)::value>(u[C]); });
          static_for<C + 2, 16>([&](auto kk) {
            constexpr int K = decltype(kk)::value;
            u[K] = __builtin_fma(nf, sk[K], u[K]);
          });
          static_for<C + 2, 16>([&](auto kk) {
            constexpr int K = decltype(kk)::value;
            x[K] = __builtin_fma(ng, sk[K], x[K]);
          });
        });
        if (bad != 0 && lane == 0) atomicCAS(info, 0, row0 + c0 + bad);   // not positive definite (or NaN)
        double pv = 1.0;                         // lane i: its own pivot d_i
#pragma unroll
        for (int c = 0; c < 16; ++c) pv = (i == c) ? u[c] : pv;
        const double rs = 1.0 / __builtin_sqrt(pv);
        static_for<0, 16>([&](auto cc) {
          constexpr int C = decltype(cc)::value;
          const double sc = row_bcast<C>(rs);
          u[C] *= sc;
          x[C] *= sc;
        });
      } else {
        // the block already holds L_jj: only the identity rows are solved, x_c /= L_cc, x_k -= x_c L_kc
        static_for<0, 16>([&](auto cc) {
          constexpr int C = decltype(cc)::value;
          const double xc = x[C] / row_bcast<C>(u[C]);
          x[C] = xc;
          static_for<C + 1, 16>([&](auto kk) {
            constexpr int K = decltype(kk)::value;
            x[K] = __builtin_fma(-xc, row_bcast<K>(u[C]), x[K]);
          });
        });
      }
      if (lane < 16 && FACTOR) {
#pragma unroll
        for (int c = 0; c < 16; ++c)
          if (c <= i) a[(c0 + i) * LSA + c0 + c] = u[c];
      }
      if (lane >= 16 && lane < 32) {
#pragma unroll
        for (int c = 0; c < 16; ++c) wd[(jb * 16 + i) * LSW + c] = x[c];
      }
    }
    __syncthreads();
    // ---- P3: the panel below the diagonal block, X = A W_jj^T (the solve x L_jj^T = a through the explicit
    // inverse), one 16 x 16 block per wave on the MFMA
    if (FACTOR) {
      for (int ib = jb + 1 + wave; ib < 8; ib += 4) {
        d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          const double av = a[(16 * ib + lr) * LSA + c0 + 4 * s4 + lq];          // A[r][k]
          const double bv = wd[(jb * 16 + 4 * s4 + lq) * LSW + lr];              // W_jj[c][k] = wd[jb][k][c]
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) a[(16 * ib + lq + 4 * r) * LSA + c0 + lr] = acc[r];
      }
      __syncthreads();
    }
  }

  // ---- inverse, block diagonal by block diagonal
  for (int d = 1; d < 8; ++d) {
    for (int i = d + wave; i < 8; i += 4) {
      const int j = i - d;
      d4 S = {0.0, 0.0, 0.0, 0.0};
      for (int k = j; k < i; ++k) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const double av = a[(16 * i + lr) * LSA + 16 * k + 4 * s + lq];                 // L_ik[r][k']
          const double bv = (k == j) ? wd[(j * 16 + lr) * LSW + 4 * s + lq]               // W_jj[k'][c]
                                     : a[(16 * j + lr) * LSA + 16 * k + 4 * s + lq];       // W_kj[k'][c]
          S = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, S, 0, 0, 0);
        }
      }
      d4 R = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const double av = wd[(i * 16 + 4 * s + lq) * LSW + lr];                           // W_ii[r][k']
        R = __builtin_amdgcn_mfma_f64_16x16x4f64(av, S[s], R, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) a[(16 * j + lr) * LSA + 16 * i + lq + 4 * r] = -R[r];    // W_ij^T
    }
    __syncthreads();
  }

#pragma unroll 8
  for (int u = 0; u < 32; ++u) {
    const int e = tid + 256 * u, i = e >> 6, j = (e & 63) * 2;        // entries (i, j) and (i, j + 1), j even
    if (FACTOR) {                                                        // the factor: lower triangle only
      if (j + 1 <= i) *reinterpret_cast<dv2*>(A + (long long)i * lda + j) = *reinterpret_cast<const dv2*>(a + i * LSA + j);
      else if (j == i) A[(long long)i * lda + j] = a[i * LSA + j];
    }
    const int bi = i >> 4, bj = j >> 4, il = i & 15, jl = j & 15;
    dv2 w = {0.0, 0.0};
    if (bi > bj) {
      w.x = a[(16 * bj + jl) * LSA + 16 * bi + il];
      w.y = a[(16 * bj + jl + 1) * LSA + 16 * bi + il];
    } else if (bi == bj) {
      w.x = wd[(bi * 16 + jl) * LSW + il];
      w.y = wd[(bi * 16 + jl + 1) * LSW + il];
    }
    *reinterpret_cast<dv2*>(W + i * NB + j) = w;
  }
}

__global__ void pack_rhs_kernel(const double* __restrict__ Y, long long N, int P, double* __restrict__ Yp,
                                long long Np) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= Np * NB) return;
  const long long i = e >> 7;
  const int c = (int)(e & 127);
  Yp[e] = (i < N && c < P) ? Y[i * P + c] : 0.0;
}
__global__ void unpack_rhs_kernel(const double* __restrict__ Yp, long long N, int P, double* __restrict__ out) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N * P) return;
  const long long i = e / P;
  const int c = (int)(e - i * P);
  out[e] = Yp[i * NB + c];
}
__global__ void tril_to_f32_kernel(const double* __restrict__ L, long long Np, long long ldl,
                                   float* __restrict__ Lf, long long ldlf) {
  // row = blockIdx.x; the lower triangle is converted and the band of GPK_ZERO_BAND_TILES tiles from the diagonal
  // tile rightwards is zeroed (the K5 launch reads that far: k_super in gpk_gemm)
  const long long i = blockIdx.x;
  const long long j = (long long)blockIdx.y * 256 + threadIdx.x;
  const long long jend = min(Np, (i / NB + GPK_ZERO_BAND_TILES) * NB);
  if (j >= jend) return;
  Lf[i * ldlf + j] = (j <= i) ? (float)L[i * ldl + j] : 0.f;
}
__global__ void to_f32_kernel(const double* __restrict__ a, long long n, float* __restrict__ b) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) b[e] = (float)a[e];
}
__global__ void copy_leaf_kernel(const double* __restrict__ w0, double* __restrict__ W0, long long ldw,
                                 long long stride_w, long long stride_W, long long stride_w2, long long stride_W2) {
  const double* __restrict__ w = reinterpret_cast<const double*>(reinterpret_cast<const char*>(w0) + blockIdx.y * stride_w +
                                                                 blockIdx.z * stride_w2);
  double* __restrict__ W = reinterpret_cast<double*>(reinterpret_cast<char*>(W0) + blockIdx.y * stride_W + blockIdx.z * stride_W2);
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= NB * NB) return;
  const int i = e >> 7, j = e & 127;
  W[(long long)i * ldw + j] = w[e];
}
// ---- LML terms: sum log diag(L), sum_i y_ip alpha_ip ----------------------------------------------
__global__ __launch_bounds__(256) void lml_terms_kernel(const double* __restrict__ L, long long N, long long ldl,
                                                        const double* __restrict__ Y,
                                                        const double* __restrict__ alpha, int P,
                                                        double* __restrict__ out, const int* __restrict__ d_info,
                                                        const int* __restrict__ gave_up, int nb, int* __restrict__ status) {
  // blockIdx.x == 0: log-det term; blockIdx.x == 1 + p: quadratic term of output p
  __shared__ double red[4];
  const int tid = threadIdx.x, b = blockIdx.x;
  // (status != null: the evaluation chain's status words ride along - the pivot failures of the nb problems and the one-launch
  // factorisation's "gave up" flag next to the results, what used to be a launch of its own)
  if (status && b == 0) {
    if (tid < nb) status[tid] = d_info[tid];
    if (tid == GPK_MAX_BATCH) status[tid] = gave_up ? *gave_up : 0;
  }
  double s = 0.0;
  if (b == 0) {
    for (long long i = tid; i < N; i += 256) s += log(L[i * ldl + i]);
  } else {
    const int p = b - 1;
    for (long long i = tid; i < N; i += 256) s = __builtin_fma(Y[i * P + p], alpha[i * P + p], s);
  }
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) out[b] = red[0] + red[1] + red[2] + red[3];
}

// ---- 256-wide base case of the right-hand triangular solve, one launch:
//   [B1 B2] <- [B1 B2] inv(L)^T  for the 256 x 256 lower block L = [L11 0; L21 L22] with inverse leaves W1, W2:
//   X1 = B1 W1^T;   X2 = (B2 - X1 L21^T) W2^T.
// The recursion spent three launches here (leaf product, 128-deep update, leaf product), each of them bound by
// ONE CU's fp64 MFMA rate per 128-row tile or, for tall panels, by eight short dependent k-steps per workgroup.  Here a
// workgroup owns 32 rows (4x the workgroups) and chains the three products: every operand is read straight from
// L2 as k-contiguous 16-byte pieces (no LDS staging, as in gpk_small.hip), the intermediate 32 x 128 result
// passes through LDS to become the next product's A operand.  In place: a workgroup reads only its own rows.
constexpr int TR = 32;
typedef double dv2t __attribute__((ext_vector_type(2)));

// acc[rb][cb] += sum_k A[16 rb + i][k] * Bm[n][k] over k = 0..127, wave-private 32 x 32 block (2 x 2 MFMA blocks).
// ap: this lane's row (i) of the A operand for rb = 0 (+ 16 rows for rb = 1), already offset by 2 kq;
// bp: this lane's row n of Bm for cb = 0 (+ 16 rows for cb = 1), already offset by 2 kq.  NEG: subtract.
template <bool NEG>
__device__ __forceinline__ void chain_product(const double* ap, long long a_rs, const double* bp, long long b_rs,
                                              d4 (&acc)[2][2]) {
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {                      // two halves of k: 8 steps of 8 k each
    dv2t a[2][8], b[2][8];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        a[x][s] = *reinterpret_cast<const dv2t*>(ap + x * 16 * a_rs + 64 * hh + 8 * s);
        b[x][s] = *reinterpret_cast<const dv2t*>(bp + x * 16 * b_rs + 64 * hh + 8 * s);
      }
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          const double ax = NEG ? -a[rb][s].x : a[rb][s].x, ay = NEG ? -a[rb][s].y : a[rb][s].y;
          acc[rb][cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(ax, b[cb][s].x, acc[rb][cb], 0, 0, 0);
          acc[rb][cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(ay, b[cb][s].y, acc[rb][cb], 0, 0, 0);
        }
  }
}

__global__ __launch_bounds__(256) void trsm256_kernel(double* __restrict__ B0, long long ldb,
                                                      const double* __restrict__ L0, long long ldl,
                                                      const double* __restrict__ W0, long long strideB,
                                                      long long strideL, long long strideW) {
  constexpr int XS = 130;                               // LDS row stride: conflict-free 16-byte operand reads
  __shared__ __attribute__((aligned(16))) double xs[TR * XS];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i = lane & 15, kq = lane >> 4;
  double* B = reinterpret_cast<double*>(reinterpret_cast<char*>(B0) + blockIdx.y * strideB) + (long long)blockIdx.x * TR * ldb;
  const double* L = reinterpret_cast<const double*>(reinterpret_cast<const char*>(L0) + blockIdx.y * strideL);
  const double* W = reinterpret_cast<const double*>(reinterpret_cast<const char*>(W0) + blockIdx.y * strideW);
  const int c0 = 32 * w;                                // this wave's 32 output columns
  d4 acc[2][2];
  // accumulator map: column = lane & 15, row = (lane >> 4) + 4 reg
  auto zero = [&]() {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) acc[rb][cb] = d4{0.0, 0.0, 0.0, 0.0};
  };
  auto to_lds = [&]() {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) xs[(16 * rb + kq + 4 * r) * XS + c0 + 16 * cb + i] = acc[rb][cb][r];
  };
  // X1 = B1 W1^T
  zero();
  chain_product<false>(B + (long long)i * ldb + 2 * kq, ldb, W + (long long)(c0 + i) * NB + 2 * kq, NB, acc);
  __syncthreads();                                      // every wave has read B1 (all 128 columns) before it changes
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) B[(long long)(16 * rb + kq + 4 * r) * ldb + c0 + 16 * cb + i] = acc[rb][cb][r];
  to_lds();
  __syncthreads();
  // T = B2 - X1 L21^T
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[rb][cb][r] = B[(long long)(16 * rb + kq + 4 * r) * ldb + NB + c0 + 16 * cb + i];
  chain_product<true>(xs + i * XS + 2 * kq, XS, L + (long long)(NB + c0 + i) * ldl + 2 * kq, ldl, acc);
  __syncthreads();                                      // X1 has been consumed: the buffer takes T
  to_lds();
  __syncthreads();
  // X2 = T W2^T
  zero();
  chain_product<false>(xs + i * XS + 2 * kq, XS, W + (long long)(NB + c0 + i) * NB + 2 * kq, NB, acc);
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) B[(long long)(16 * rb + kq + 4 * r) * ldb + NB + c0 + 16 * cb + i] = acc[rb][cb][r];
}

// ---- K3 at large N: alpha = W^T (W Y) with a handful of right-hand sides, as two streaming passes over W ------
// With P <= 6 targets the GEMM form of gpk_potrs_inv pads the right-hand sides to a 128-column panel and runs W
// through 128 x 128 MFMA tiles: 2 x 17 GB at 1.8 TB/s at N = 65 536.  The work is a matrix-vector product per target:
// bound by reading W once per pass.  These kernels stream the lower triangle as 1 KiB row segments (one wave
// instruction = 128 consecutive doubles of one row, 8 or 16 of them in flight per wave) and keep the right-hand sides in
// registers; every reduction runs in a fixed order (bit-reproducible, no atomics).
//   pass 1  Z = W Y:      a workgroup owns RB rows, its 4 waves take the 128-column chunks c = wave, wave + 4, ...;
//                         a lane keeps y of its two columns (2 P doubles) and RB x P partial sums, reduced over the
//                         lanes and waves once per row block
//   pass 2  alpha = W^T Z: a workgroup owns (128-column chunk, segment of SEG rows); a wave reads whole row segments
//                         (rows wave, wave + 4, ...), z_i is wave-uniform; partial column sums per segment, then a
//                         reduction over the segments. SEG follows the size (k3_seg): 2048 rows from Np = 32768,
//                         down to 128 at Np <= 4096, so that the launch has about 1000 workgroups at every size
//                         (with 2048 rows, Np = 4096 was 48 workgroups of 512 rows per wave: 120 us for 67 MB)
inline int k3_seg(int64_t Np) {
  int seg = 128;
  while (seg < 2048 && (long long)seg * 2 * 262144 <= (long long)Np * Np) seg *= 2;
  return seg;
}
template <int P, int RB>
__global__ __launch_bounds__(256) void k3_wy_kernel(const double* __restrict__ W, long long ldw, const double* __restrict__ Y,
                                                    long long N, double* __restrict__ Z, int nblk) {
  typedef double dv2 __attribute__((ext_vector_type(2)));
  __shared__ double red[4][RB * P];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rb = nblk - 1 - (int)blockIdx.x;                  // heavy first: the longest rows start first
  const long long r0 = (long long)rb * RB;
  const int nch = (int)((r0 + RB - 1) / 128) + 1;             // 128-column chunks that reach this row block
  double acc[RB][P];
#pragma unroll
  for (int r = 0; r < RB; ++r)
#pragma unroll
    for (int p = 0; p < P; ++p) acc[r][p] = 0.0;
  for (int c = wave; c < nch; c += 4) {
    const long long j = (long long)c * 128 + 2 * lane;
    double y0[P], y1[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      y0[p] = j < N ? Y[j * P + p] : 0.0;
      y1[p] = j + 1 < N ? Y[(j + 1) * P + p] : 0.0;
    }
    dv2 w[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) w[r] = *reinterpret_cast<const dv2*>(W + (r0 + r) * ldw + j);
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      // (entries right of the diagonal are not part of W: they exist only inside the diagonal chunk)
      const double wx = j <= r0 + r ? w[r].x : 0.0, wy = j + 1 <= r0 + r ? w[r].y : 0.0;
#pragma unroll
      for (int p = 0; p < P; ++p) acc[r][p] = __builtin_fma(wy, y1[p], __builtin_fma(wx, y0[p], acc[r][p]));
    }
  }
#pragma unroll
  for (int r = 0; r < RB; ++r)
#pragma unroll
    for (int p = 0; p < P; ++p) {
      double v = acc[r][p];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
      if (lane == 0) red[wave][r * P + p] = v;
    }
  __syncthreads();
  if (tid < RB * P) Z[r0 * P + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

template <int P>
__global__ __launch_bounds__(256) void k3_wtz_kernel(const double* __restrict__ W, long long ldw, long long Np,
                                                     const double* __restrict__ Z, double* __restrict__ part, int nseg, int K3_SEG) {
  typedef double dv2 __attribute__((ext_vector_type(2)));
  __shared__ double red[4][128 * P];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // item -> (column chunk c, row segment s >= first segment that reaches the chunk); enumerated chunk-major, the
  // chunks with the most segments first
  const int c = (int)blockIdx.y, s = (int)blockIdx.x;
  const long long j0 = (long long)c * 128, i_lo = (long long)s * K3_SEG, i_hi = min(Np, i_lo + K3_SEG);
  if (i_hi <= j0) return;                                     // the segment lies above the chunk's diagonal
  const long long j = j0 + 2 * lane;
  double a0[P], a1[P];
#pragma unroll
  for (int p = 0; p < P; ++p) a0[p] = a1[p] = 0.0;
  const long long ib = max(i_lo, j0);
  constexpr int U = 8;
  for (long long i = ib + wave; i < i_hi; i += 4 * U) {
    dv2 w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long ii = i + 4 * u;
      w[u] = ii < i_hi ? *reinterpret_cast<const dv2*>(W + ii * ldw + j) : dv2{0.0, 0.0};
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long ii = __builtin_amdgcn_readfirstlane((int)min(i + 4 * u, i_hi - 1));   // wave-uniform row
      const double wx = j <= i + 4 * u ? w[u].x : 0.0, wy = j + 1 <= i + 4 * u ? w[u].y : 0.0;
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const double z = Z[ii * P + p];
        a0[p] = __builtin_fma(wx, z, a0[p]);
        a1[p] = __builtin_fma(wy, z, a1[p]);
      }
    }
  }
#pragma unroll
  for (int p = 0; p < P; ++p) {
    red[wave][(2 * lane) * P + p] = a0[p];
    red[wave][(2 * lane + 1) * P + p] = a1[p];
  }
  __syncthreads();
  for (int e = tid; e < 128 * P; e += 256)
    part[((long long)s * Np + j0) * P + e] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
}

template <int P>
__global__ void k3_reduce_kernel(const double* __restrict__ part, long long Np, long long N, int nseg, int K3_SEG,
                                 double* __restrict__ alpha) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= N * P) return;
  const long long jcol = e / P;
  double v = 0.0;
  for (int s = (int)((jcol / 128 * 128) / K3_SEG); s < nseg; ++s) v += part[(long long)s * Np * P + e];
  alpha[e] = v;
}

template <int P>
int k3_stream(gpk_handle h, const double* W, int64_t Np, int64_t ldw, const double* Y, int64_t N, double* alpha) {
  constexpr int RB = P <= 3 ? 16 : 8;
  const int K3_SEG = k3_seg(Np), nseg = (int)((Np + K3_SEG - 1) / K3_SEG);
  void* ws = nullptr;
  GPK_TRY(gpk_scratch(h, ((size_t)Np * P + (size_t)nseg * Np * P) * sizeof(double), &ws));
  double* Z = (double*)ws;
  double* part = Z + Np * P;
  const int nblk = (int)(Np / RB);
  hipLaunchKernelGGL((k3_wy_kernel<P, RB>), dim3((unsigned)nblk), dim3(256), 0, h->stream, W, (long long)ldw, Y, (long long)N, Z, nblk);
  GPK_LAUNCH_CHECK(h);
  hipLaunchKernelGGL((k3_wtz_kernel<P>), dim3((unsigned)nseg, (unsigned)(Np / 128)), dim3(256), 0, h->stream, W, (long long)ldw,
                     (long long)Np, (const double*)Z, part, nseg, K3_SEG);
  GPK_LAUNCH_CHECK(h);
  hipLaunchKernelGGL((k3_reduce_kernel<P>), dim3((unsigned)((N * P + 255) / 256)), dim3(256), 0, h->stream, (const double*)part,
                     (long long)Np, (long long)N, nseg, K3_SEG, alpha);
  GPK_LAUNCH_CHECK(h);
  return GPK_OK;
}

inline int half_split(int64_t n) { return (int)(((n / NB) / 2) * NB); }

template <typename T> constexpr int dt();
template <> constexpr int dt<double>() { return GPK_F64; }
template <> constexpr int dt<float>() { return GPK_F32; }

// B (m x n) <- B * L^-T, L (n x n) lower; winv = inverse leaves of L (n x 128)
int trsm_right_rec(gpk_handle h, double* B, int64_t ldb, int64_t m, const double* L, int64_t ldl, int64_t n,
                   const double* winv) {
  if (n == NB) {
    GemmArgs g = gemm_args(B, ldb, 0, winv, NB, 0, B, ldb, (int)m, NB, NB, 1.0, 0.0);
    return gpk_gemm(h, GPK_F64, g);
  }
  if (n == 2 * NB && h->trsm256) {                      // two leaves and the update between them: one launch
    hipLaunchKernelGGL(trsm256_kernel, dim3((unsigned)(m / TR), h->batch), dim3(256), 0, h->stream, B, (long long)ldb, L,
                       (long long)ldl, winv, gpk_bstride(h, B), gpk_bstride(h, L), gpk_bstride(h, winv));
    GPK_LAUNCH_CHECK(h);
    return GPK_OK;
  }
  const int64_t n1 = half_split(n), n2 = n - n1;
  GPK_TRY(trsm_right_rec(h, B, ldb, m, L, ldl, n1, winv));
  // B2 -= B1 * L21^T
  GemmArgs g = gemm_args(B, ldb, 0, L + n1 * ldl, ldl, 0, B + n1, ldb, (int)m, (int)n2, (int)n1, -1.0, 1.0);
  GPK_TRY(gpk_gemm(h, GPK_F64, g));
  return trsm_right_rec(h, B + n1, ldb, m, L + n1 * ldl + n1, ldl, n2, winv + n1 * NB);
}

int potrf_rec(gpk_handle h, double* A, int64_t lda, int64_t n, double* winv, int64_t row0) {
  if (n <= h->ptile_max_np && n >= 4 * NB) {
    // the bottom of the recursion: one persistent launch per diagonal block of this size (gpk_ptile.hip) instead of its
    // ~3 n / 128 dependent launches
    int used = 0;
    GPK_TRY(gpk_potrf_ptile(h, A, n, lda, winv, (int)row0, &used));
    if (used) return GPK_OK;
  }
  if (n == NB) {
    hipLaunchKernelGGL(leaf_kernel<true>, dim3(h->batch), dim3(256), 0, h->stream, A, (long long)lda, (int)row0,
                       h->d_info, winv, gpk_bstride(h, A), gpk_bstride(h, winv));
    GPK_LAUNCH_CHECK(h);
    return GPK_OK;
  }
  const int64_t n1 = half_split(n), n2 = n - n1;
  GPK_TRY(potrf_rec(h, A, lda, n1, winv, row0));
  double* A21 = A + n1 * lda;
  double* A22 = A21 + n1;
  GPK_TRY(trsm_right_rec(h, A21, lda, n2, A, lda, n1, winv));
  GemmArgs g = gemm_args(A21, lda, 0, A21, lda, 0, A22, lda, (int)n2, (int)n2, (int)n1, -1.0, 1.0);
  g.lower_only = 1;
  GPK_TRY(gpk_gemm(h, GPK_F64, g));
  return potrf_rec(h, A22, lda, n2, winv + n1 * NB, row0 + n1);
}

// B (n x M) <- L^-1 B
template <typename T>
int trsm_left_rec(gpk_handle h, T* B, int64_t ldb, int64_t M, const T* L, int64_t ldl, int64_t n, const T* winv) {
  if (n == NB) {
    GemmArgs g = gemm_args(winv, NB, 0, B, ldb, 1, B, ldb, NB, (int)M, NB, 1.0, 0.0);
    return gpk_gemm(h, dt<T>(), g);
  }
  const int64_t n1 = half_split(n), n2 = n - n1;
  GPK_TRY(trsm_left_rec<T>(h, B, ldb, M, L, ldl, n1, winv));
  // B2 -= L21 * B1
  GemmArgs g = gemm_args(L + n1 * ldl, ldl, 0, B, ldb, 1, B + n1 * ldb, ldb, (int)n2, (int)M, (int)n1, -1.0, 1.0);
  GPK_TRY(gpk_gemm(h, dt<T>(), g));
  return trsm_left_rec<T>(h, B + n1 * ldb, ldb, M, L + n1 * ldl + n1, ldl, n2, winv + n1 * NB);
}

// B (n x M) <- L^-T B
int trsm_left_t_rec(gpk_handle h, double* B, int64_t ldb, int64_t M, const double* L, int64_t ldl, int64_t n,
                    const double* winv) {
  if (n == NB) {
    GemmArgs g = gemm_args(winv, NB, 1, B, ldb, 1, B, ldb, NB, (int)M, NB, 1.0, 0.0);
    return gpk_gemm(h, GPK_F64, g);
  }
  const int64_t n1 = half_split(n), n2 = n - n1;
  GPK_TRY(trsm_left_t_rec(h, B + n1 * ldb, ldb, M, L + n1 * ldl + n1, ldl, n2, winv + n1 * NB));
  // B1 -= L21^T * B2   (L21 stored n2 x n1 = k x m)
  GemmArgs g = gemm_args(L + n1 * ldl, ldl, 1, B + n1 * ldb, ldb, 1, B, ldb, (int)n1, (int)M, (int)n2, -1.0, 1.0);
  GPK_TRY(gpk_gemm(h, GPK_F64, g));
  return trsm_left_t_rec(h, B, ldb, M, L, ldl, n1, winv);
}

// W (n x n, ldw) <- L^-1 (lower); T: scratch at least (n/2) x (n/2) with leading dimension ldt
int trtri_rec(gpk_handle h, const double* L, int64_t ldl, int64_t n, const double* winv, double* W, int64_t ldw,
              double* T, int64_t ldt) {
  if (n == NB) {
    hipLaunchKernelGGL(copy_leaf_kernel, dim3(NB * NB / 256, h->batch), dim3(256), 0, h->stream, winv, W,
                       (long long)ldw, gpk_bstride(h, winv), gpk_bstride(h, W), 0ll, 0ll);
    GPK_LAUNCH_CHECK(h);
    return GPK_OK;
  }
  const int64_t n1 = half_split(n), n2 = n - n1;
  GPK_TRY(trtri_rec(h, L, ldl, n1, winv, W, ldw, T, ldt));
  GPK_TRY(trtri_rec(h, L + n1 * ldl + n1, ldl, n2, winv + n1 * NB, W + n1 * ldw + n1, ldw, T, ldt));
  // T (n2 x n1) = L21 * W11   (W11 lower: k >= column tile start)
  GemmArgs g = gemm_args(L + n1 * ldl, ldl, 0, W, ldw, 1, T, ldt, (int)n2, (int)n1, (int)n1, 1.0, 0.0);
  g.kb_col = NB;
  g.k_super = 1;           // W is zero right of the diagonal for GPK_ZERO_BAND_TILES - 1 tiles (zero_band_kernel)
  GPK_TRY(gpk_gemm(h, GPK_F64, g));
  // W21 = -W22 * T          (W22 lower: k < row tile end)
  GemmArgs g2 = gemm_args(W + n1 * ldw + n1, ldw, 0, T, ldt, 1, W + n1 * ldw, ldw, (int)n2, (int)n1, (int)n2, -1.0, 0.0);
  g2.ke0 = NB; g2.ke_row = NB;
  g2.k_super = 1;
  g2.heavy_first = 1;      // row tile r costs r + 1 k-blocks: longest first
  return gpk_gemm(h, GPK_F64, g2);
}

// Level by level instead of depth first: the diagonal blocks of one level are independent, so each level is ONE batched
// launch per product (and the leaves one copy launch) instead of Np / n launches of a few tiles each.  Blocks of
// n = 256, 512, ... rows are formed bottom-up from pairs of finished n / 2 blocks; when the tile count is not a power of
// two the rows behind the last whole block form a ragged tail block that is merged with a whole n / 2 block whenever the
// tail grows past n / 2 (one more pair of launches at that level).  2 log2(Np / 128) + 1 launches for a power of two
// (N = 4096: 11 instead of 94), at most twice that otherwise (N = 10 000, 79 tiles: 23 instead of 158).  With the
// handle in batched mode every launch covers all problems of the batch.
// T: scratch of at least (Np / 2 + 128)^2 doubles (per problem); a launch lays it out with its own leading dimension.
// max |(float)w_ij| over the lower triangle of the diagonal tiles (the tile inverses), per tile: the part of a 128-row
// block's maximum that no product writes
__global__ __launch_bounds__(256) void leaf_absmax_kernel(const double* __restrict__ winv, unsigned* __restrict__ out) {
  const double* w = winv + (long long)blockIdx.x * NB * NB;
  float m = 0.f;
  for (int e = threadIdx.x; e < NB * NB; e += 256)
    if ((e & 127) <= (e >> 7)) m = fmaxf(m, fabsf((float)w[e]));
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(out + blockIdx.x, __float_as_uint(m));
}

// amax != nullptr (one problem only): max |(float)W_ij| per 128-row block of the lower triangle is accumulated there as the
// tiles are written (the products' epilogue; the diagonal tiles by leaf_absmax_kernel) - the first pass of
// gpk_split2_rows_f64 without reading W again
int trtri_levels(gpk_handle h, const double* L, int64_t ldl, int64_t Np, const double* winv, double* W, int64_t ldw,
                 double* T, unsigned* amax = nullptr) {
  const int64_t nl = Np / NB;
  if (amax) {
    GPK_CHECK_HIP(h, hipMemsetAsync(amax, 0, nl * sizeof(unsigned), h->stream));
    hipLaunchKernelGGL(leaf_absmax_kernel, dim3((unsigned)nl), dim3(256), 0, h->stream, winv, amax);
    GPK_LAUNCH_CHECK(h);
  }
  hipLaunchKernelGGL(copy_leaf_kernel, dim3(NB * NB / 256, (unsigned)nl, (unsigned)h->batch), dim3(256), 0, h->stream, winv, W,
                     (long long)ldw, (long long)(NB * NB * sizeof(double)), (long long)(NB * (ldw + 1) * sizeof(double)),
                     gpk_bstride(h, winv), gpk_bstride(h, W));
  GPK_LAUNCH_CHECK(h);
  // one merge: rows [r0, r0 + mt) x columns [c0, c0 + h2) of W from the finished blocks W11 = W[c0.., c0..] (h2 x h2)
  // and W22 = W[r0.., r0..] (mt x mt), r0 = c0 + h2; `nb` such merges n rows apart
  auto merge = [&](int64_t c0, int64_t h2, int64_t mt, int64_t n, int64_t nb) -> int {
    const int64_t r0 = c0 + h2, ldt = nb * h2;
    // T_b (mt x h2) = L21_b * W11_b   (W11 lower: k >= column tile start)
    GemmArgs g = gemm_args(L + r0 * ldl + c0, ldl, 0, W + c0 * ldw + c0, ldw, 1, T, ldt, (int)mt, (int)h2, (int)h2, 1.0, 0.0);
    g.kb_col = NB;
    g.k_super = 1;
    g.nbatch = (int)nb;
    g.sA = (long long)(n * (ldl + 1) * sizeof(double));
    g.sB = (long long)(n * (ldw + 1) * sizeof(double));
    g.sC = (long long)(h2 * sizeof(double));             // block b's scratch: columns [b h2, (b + 1) h2) of T
    GPK_TRY(gpk_gemm(h, GPK_F64, g));
    // W21_b = -W22_b * T_b            (W22 lower: k < row tile end)
    GemmArgs g2 = gemm_args(W + r0 * ldw + r0, ldw, 0, T, ldt, 1, W + r0 * ldw + c0, ldw, (int)mt, (int)h2, (int)mt, -1.0, 0.0);
    g2.ke0 = NB; g2.ke_row = NB;
    g2.k_super = 1;
    g2.heavy_first = 1;
    g2.nbatch = (int)nb;
    g2.sA = (long long)(n * (ldw + 1) * sizeof(double));
    g2.sB = (long long)(h2 * sizeof(double));
    g2.sC = (long long)(n * (ldw + 1) * sizeof(double));
    if (amax) { g2.epilogue = 2; g2.amax = amax; g2.amax_base = W; }
    return gpk_gemm(h, GPK_F64, g2);
  };
  for (int64_t n = 2 * NB; n / 2 < Np; n *= 2) {
    const int64_t h2 = n / 2, nb = Np / n, rem = Np - nb * n;
    if (nb > 0) GPK_TRY(merge(0, h2, h2, n, nb));
    if (rem > h2) GPK_TRY(merge(nb * n, h2, rem - h2, n, 1));     // the tail: a whole n / 2 block and what was behind it
  }
  return GPK_OK;
}

}  // namespace

namespace {
__global__ void zero_band_kernel(double* __restrict__ W0, long long Np, long long ldw, long long strideW);   // (defined with gpk_trtri)
// W (lower tiles, strictly below the block diagonal) from W^T (upper tiles): 64 x 64 pieces transposed through LDS
__global__ __launch_bounds__(256) void mirror_lower_kernel(const double* __restrict__ Wt0, long long ldt, double* __restrict__ W0,
                                                           long long ldw, int nt, long long strideWt, long long strideW,
                                                           const double* __restrict__ winv0, long long strideWinv) {
  __shared__ double t[64][65];
  const double* __restrict__ Wt = reinterpret_cast<const double*>(reinterpret_cast<const char*>(Wt0) + blockIdx.y * strideWt);
  double* __restrict__ W = reinterpret_cast<double*>(reinterpret_cast<char*>(W0) + blockIdx.y * strideW);
  // blockIdx.x: piece (a, b), a, b in 0..1, of tile pair p = (i, j), i > j, enumerated row by row; behind them the four pieces
  // of every diagonal tile, copied as they are from the tile inverses (copy_leaf_kernel's job, in the same launch)
  const int piece = blockIdx.x & 3, pr = blockIdx.x >> 2;
  const int npair = nt * (nt - 1) / 2;
  if (pr >= npair) {
    const int d = pr - npair, a = piece >> 1, b = piece & 1;
    if (d >= nt) return;
    const double* w = reinterpret_cast<const double*>(reinterpret_cast<const char*>(winv0) + blockIdx.y * strideWinv) + (long long)d * 128 * 128;
    double* dst = W + ((long long)(128 * d + 64 * a)) * ldw + 128 * d + 64 * b;
    const int c = threadIdx.x & 63, r0 = threadIdx.x >> 6;
    for (int r = r0; r < 64; r += 4) dst[(long long)r * ldw + c] = w[(64 * a + r) * 128 + 64 * b + c];
    return;
  }
  int i = (int)((1.0f + __builtin_sqrtf(8.0f * (float)pr + 1.0f)) * 0.5f);
  while (i * (i - 1) / 2 > pr) --i;
  while ((i + 1) * i / 2 <= pr) ++i;
  const int j = pr - i * (i - 1) / 2;
  if (i >= nt) return;
  const int a = piece >> 1, b = piece & 1;
  // source: rows of tile (j, i) of W^T: rows 128 j + 64 b .., columns 128 i + 64 a ..;  destination: tile (i, j) of W
  const double* src = Wt + ((long long)(128 * j + 64 * b)) * ldt + 128 * i + 64 * a;
  double* dst = W + ((long long)(128 * i + 64 * a)) * ldw + 128 * j + 64 * b;
  const int c = threadIdx.x & 63, r0 = threadIdx.x >> 6;
  for (int r = r0; r < 64; r += 4) t[r][c] = src[(long long)r * ldt + c];
  __syncthreads();
  for (int r = r0; r < 64; r += 4) dst[(long long)r * ldw + c] = t[c][r];
}
}  // namespace

// Factor AND inverse factor as one persistent launch (gpk_ptile.hip with the tiles of W^T in its task list) + a transposing copy:
// for matrices small enough that the factorisation leaves most of the chip idle (h->ptile_inv_max_np).  wt: Np x lda scratch.
// *used = 0: not served (the caller runs gpk_potrf_enqueue + gpk_trtri).
int gpk_potrf_trtri_enqueue(gpk_handle h, double* A, int64_t Np, int64_t lda, double* winv, double* W, int64_t ldw, double* wt,
                            int* used) {
  *used = 0;
  // (a batch fills the chip with its factorisation tasks: N = 4096 x 3 measured 5.38 ms with the tiles of W^T in the launch,
  // 5.27 with the level products - the bound is on the rows of the whole batch)
  if (!wt || Np * h->batch > h->ptile_inv_max_np || ldw != lda || ((uintptr_t)wt % 128) != 0 || (gpk_bstride(h, wt) % 128) != 0)
    return GPK_OK;
  if (h->batch > 1 && (gpk_bstride(h, wt) == 0 || gpk_bstride(h, W) == 0)) return GPK_OK;   // (every problem needs its own W^T and W)
  GPK_REQUIRE(h, A && winv && W, "potrf: null pointer");
  gpk_time_begin(h, GPK_TIMED_POTRF);
  h->ptile_launches = 0;
  // (the launch's own preparation also zeroes the pivot words and the band right of W's diagonal tiles: one launch for what were
  // three memsets and two kernels)
  const int rc = gpk_potrf_ptile(h, A, Np, lda, winv, 0, used, wt, W, ldw, 1);
  gpk_time_end(h);
  GPK_TRY(rc);
  if (!*used) return GPK_OK;
  const int64_t nl = Np / NB;
  // the diagonal tiles of W from the tile inverses, the rest mirrored: one launch
  hipLaunchKernelGGL(mirror_lower_kernel, dim3((unsigned)((nl * (nl - 1) / 2 + nl) * 4), (unsigned)h->batch), dim3(256), 0, h->stream,
                     wt, (long long)lda, W, (long long)ldw, (int)nl, gpk_bstride(h, wt), gpk_bstride(h, W), winv, gpk_bstride(h, winv));
  GPK_LAUNCH_CHECK(h);
  return GPK_OK;
}

// The launches of gpk_potrf without its synchronisation (gpk_lml_eval queues the rest of an evaluation behind them);
// gpk_potrf_finish reads the pivot failures back once the stream has been synchronised.
int gpk_potrf_enqueue(gpk_handle h, double* A, int64_t Np, int64_t lda, double* winv) {
  GPK_REQUIRE(h, A && winv, "potrf: null pointer");
  GPK_REQUIRE(h, Np >= NB && Np % NB == 0 && lda >= Np && lda % 2 == 0, "potrf: Np must be a positive multiple of 128");
  GPK_REQUIRE(h, Np < (1ll << 31), "potrf: Np too large");
  GPK_REQUIRE(h, ((uintptr_t)A % 16) == 0 && ((uintptr_t)winv % 16) == 0, "potrf: A, winv must be 16-byte aligned");
  GPK_CHECK_HIP(h, hipMemsetAsync(h->d_info, 0, h->batch * sizeof(int), h->stream));
  gpk_time_begin(h, GPK_TIMED_POTRF);
  h->ptile_launches = 0;
  const int rc = potrf_rec(h, A, lda, Np, winv, 0);
  gpk_time_end(h);
  return rc;
}

// after the stream has been synchronised and d_info copied to hinfo_all (one entry per problem of the batch)
int gpk_potrf_finish(gpk_handle h, const int* hinfo_all, int* info, int gave_up) {
  const int nb = h->batch;
  if (h->ptile_launches > 0) GPK_TRY(gpk_potrf_ptile_check(h, gave_up));
  int hinfo = 0;
  for (int b = 0; b < nb; ++b) {
    info[b] = hinfo_all[b];
    if (hinfo == 0 && hinfo_all[b] != 0) hinfo = hinfo_all[b];
  }
  if (hinfo != 0) {
    char buf[160];
    snprintf(buf, sizeof buf, "matrix is not positive definite: leading minor of order %d has a non-positive pivot", hinfo);
    h->err = buf;
    return GPK_NOT_PD;
  }
  return GPK_OK;
}

extern "C" int gpk_potrf(gpk_handle h, double* A, int64_t Np, int64_t lda, double* winv, int* info) {
  if (!h) return GPK_BAD_ARG;
  GPK_REQUIRE(h, info, "potrf: null pointer");
  GPK_TRY(gpk_potrf_enqueue(h, A, Np, lda, winv));
  int hinfo_all[GPK_MAX_BATCH] = {0};                // batched mode: info receives one entry per problem
  GPK_CHECK_HIP(h, hipMemcpyAsync(hinfo_all, h->d_info, h->batch * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  GPK_CHECK_HIP(h, hipStreamSynchronize(h->stream));
  return gpk_potrf_finish(h, hinfo_all, info);
}

extern "C" int gpk_leaf_inverses(gpk_handle h, const double* L, int64_t Np, int64_t ldl, double* winv) {
  if (!h) return GPK_BAD_ARG;
  GPK_REQUIRE(h, L && winv, "leaf_inverses: null pointer");
  GPK_REQUIRE(h, Np >= NB && Np % NB == 0 && ldl >= Np, "leaf_inverses: Np must be a positive multiple of 128");
  GPK_REQUIRE(h, ldl % 2 == 0 && ((uintptr_t)L % 16) == 0 && ((uintptr_t)winv % 16) == 0, "leaf_inverses: L, winv must be 16-byte aligned, ldl even");
  for (int64_t b = 0; b < Np / NB; ++b) {
    hipLaunchKernelGGL(leaf_kernel<false>, dim3(h->batch), dim3(256), 0, h->stream,
                       const_cast<double*>(L) + b * NB * ldl + b * NB, (long long)ldl, 0, h->d_info,
                       winv + b * NB * NB, gpk_bstride(h, L), gpk_bstride(h, winv));
    GPK_LAUNCH_CHECK(h);
  }
  return GPK_OK;
}

extern "C" int gpk_factor_to_f32(gpk_handle h, const double* L, int64_t Np, int64_t ldl, const double* winv,
                                 float* Lf, int64_t ldlf, float* winvf) {
  if (!h) return GPK_BAD_ARG;
  GPK_REQUIRE(h, L && winv && Lf && winvf, "factor_to_f32: null pointer");
  GPK_REQUIRE(h, Np % NB == 0 && Np > 0 && ldl >= Np && ldlf >= Np, "factor_to_f32: bad size");
  hipLaunchKernelGGL(tril_to_f32_kernel, dim3((unsigned)Np, (unsigned)((Np + 255) / 256)), dim3(256), 0, h->stream, L,
                     (long long)Np, (long long)ldl, Lf, (long long)ldlf);
  GPK_LAUNCH_CHECK(h);
  const long long n = Np * NB;
  hipLaunchKernelGGL(to_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, winv, n, winvf);
  GPK_LAUNCH_CHECK(h);
  return GPK_OK;
}

extern "C" int gpk_potrs(gpk_handle h, const double* L, int64_t Np, int64_t ldl, const double* winv,
                         const double* Y, int64_t N, int P, double* alpha) {
  if (!h) return GPK_BAD_ARG;
  GPK_REQUIRE(h, L && winv && Y && alpha, "potrs: null pointer");
  GPK_REQUIRE(h, Np % NB == 0 && N >= 1 && N <= Np && ldl >= Np, "potrs: bad sizes");
  GPK_REQUIRE(h, P >= 1 && P <= GPK_MAX_P, "potrs: P must be in [1, 16]");
  void* ws = nullptr;
  GPK_TRY(gpk_scratch(h, (size_t)Np * NB * sizeof(double), &ws));
  double* Yp = (double*)ws;
  const long long tot = Np * NB;
  hipLaunchKernelGGL(pack_rhs_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, Y,
                     (long long)N, P, Yp, (long long)Np);
  GPK_LAUNCH_CHECK(h);
  GPK_TRY(trsm_left_rec<double>(h, Yp, NB, NB, L, ldl, Np, winv));
  GPK_TRY(trsm_left_t_rec(h, Yp, NB, NB, L, ldl, Np, winv));
  hipLaunchKernelGGL(unpack_rhs_kernel, dim3((unsigned)((N * P + 255) / 256)), dim3(256), 0, h->stream,
                     (const double*)Yp, (long long)N, P, alpha);
  GPK_LAUNCH_CHECK(h);
  return GPK_OK;
}

extern "C" int gpk_trsm_lower_left(gpk_handle h, int dtype, const void* L, int64_t Np, int64_t ldl,
                                   const void* winv, void* B, int64_t Mp, int64_t ldb) {
  if (!h) return GPK_BAD_ARG;
  GPK_REQUIRE(h, L && winv && B, "trsm: null pointer");
  GPK_REQUIRE(h, Np % NB == 0 && Mp % NB == 0 && Np > 0 && Mp > 0 && ldl >= Np && ldb >= Mp, "trsm: sizes must be multiples of 128");
  if (dtype == GPK_F64)
    return trsm_left_rec<double>(h, (double*)B, ldb, Mp, (const double*)L, ldl, Np, (const double*)winv);
  GPK_REQUIRE(h, dtype == GPK_F32, "trsm: bad dtype");
  return trsm_left_rec<float>(h, (float*)B, ldb, Mp, (const float*)L, ldl, Np, (const float*)winv);
}

// the launch of gpk_lml_terms: 1 + P doubles to dout (device)
int gpk_lml_terms_enqueue(gpk_handle h, const double* L, int64_t N, int64_t ldl, const double* Y, const double* alpha, int P,
                          double* dout, int with_status) {
  GPK_REQUIRE(h, L && Y && alpha, "lml_terms: null pointer");
  GPK_REQUIRE(h, P >= 1 && P <= GPK_MAX_P && N >= 1, "lml_terms: bad sizes");
  hipLaunchKernelGGL(lml_terms_kernel, dim3(1 + P), dim3(256), 0, h->stream, L, (long long)N, (long long)ldl, Y,
                     alpha, P, dout, (const int*)h->d_info,
                     h->ptile_launches > 0 ? (const int*)(h->d_ptile + GPK_PTILE_CTRL_INTS) : (const int*)nullptr, h->batch,
                     with_status ? reinterpret_cast<int*>(h->d_small + GPK_STATUS_OFF) : (int*)nullptr);
  GPK_LAUNCH_CHECK(h);
  return GPK_OK;
}

extern "C" int gpk_lml_terms(gpk_handle h, const double* L, int64_t N, int64_t ldl, const double* Y,
                             const double* alpha, int P, double* terms) {
  if (!h) return GPK_BAD_ARG;
  GPK_REQUIRE(h, terms, "lml_terms: null pointer");
  GPK_TRY(gpk_lml_terms_enqueue(h, L, N, ldl, Y, alpha, P, h->d_small));     // (d_small is the pinned block itself: no copy)
  GPK_CHECK_HIP(h, hipStreamSynchronize(h->stream));
  for (int i = 0; i < 1 + P; ++i) terms[i] = h->h_small[i];
  return GPK_OK;
}

namespace {
// zero the tiles right of each diagonal tile of W (GPK_ZERO_BAND_TILES - 1 of them): the K5 launch gives every
// row of an 8-row super-tile the k-range of its longest row and relies on zeros beyond a row's own range
__global__ void zero_band_kernel(double* __restrict__ W0, long long Np, long long ldw, long long strideW) {
  double* __restrict__ W = reinterpret_cast<double*>(reinterpret_cast<char*>(W0) + blockIdx.z * strideW);
  const long long i = blockIdx.x;
  const long long j0 = (i / NB + 1) * NB, j1 = min(Np, (i / NB + GPK_ZERO_BAND_TILES) * NB);
  const long long j = j0 + (long long)blockIdx.y * 256 + threadIdx.x;
  if (j < j1) W[i * ldw + j] = 0.0;
}
}  // namespace

static int trtri_impl(gpk_handle h, const double* L, int64_t Np, int64_t ldl, const double* winv, double* W, int64_t ldw,
                      double* work, unsigned* amax);

extern "C" int gpk_trtri(gpk_handle h, const double* L, int64_t Np, int64_t ldl, const double* winv, double* W,
                         int64_t ldw, double* work) {
  if (!h) return GPK_BAD_ARG;
  return trtri_impl(h, L, Np, ldl, winv, W, ldw, work, nullptr);
}

extern "C" int gpk_trtri_absmax(gpk_handle h, const double* L, int64_t Np, int64_t ldl, const double* winv, double* W,
                                int64_t ldw, double* work, float* block_absmax) {
  if (!h) return GPK_BAD_ARG;
  GPK_REQUIRE(h, block_absmax, "trtri_absmax: null pointer");
  GPK_REQUIRE(h, h->batch == 1, "trtri_absmax: not available in batched mode");
  return trtri_impl(h, L, Np, ldl, winv, W, ldw, work, reinterpret_cast<unsigned*>(block_absmax));
}

static int trtri_impl(gpk_handle h, const double* L, int64_t Np, int64_t ldl, const double* winv, double* W, int64_t ldw,
                      double* work, unsigned* amax) {
  GPK_REQUIRE(h, L && winv && W && work, "trtri: null pointer");
  GPK_REQUIRE(h, Np % NB == 0 && Np > 0 && ldl >= Np && ldw >= Np, "trtri: Np must be a multiple of 128");
  if (Np > NB) {
    const unsigned ny = (unsigned)(((GPK_ZERO_BAND_TILES - 1) * NB + 255) / 256);
    hipLaunchKernelGGL(zero_band_kernel, dim3((unsigned)Np, ny, (unsigned)h->batch), dim3(256), 0, h->stream, W,
                       (long long)Np, (long long)ldw, gpk_bstride(h, W));
    GPK_LAUNCH_CHECK(h);
  }
  const int64_t n1 = half_split(Np);
  if (Np >= 2 * NB && h->trtri_levels) return trtri_levels(h, L, ldl, Np, winv, W, ldw, work, amax);
  GPK_TRY(trtri_rec(h, L, ldl, Np, winv, W, ldw, work, n1 > 0 ? n1 : NB));
  // (the depth-first form has no epilogue for it: one pass over W)
  return amax ? gpk_tril_block_absmax_f64_enqueue(h, W, Np, ldw, amax) : GPK_OK;
}

extern "C" int gpk_tril_to_f32(gpk_handle h, const double* A, int64_t Np, int64_t lda, float* Af, int64_t ldaf) {
  if (!h) return GPK_BAD_ARG;
  GPK_REQUIRE(h, A && Af, "tril_to_f32: null pointer");
  GPK_REQUIRE(h, Np % NB == 0 && Np > 0 && lda >= Np && ldaf >= Np, "tril_to_f32: bad size");
  hipLaunchKernelGGL(tril_to_f32_kernel, dim3((unsigned)Np, (unsigned)((Np + 255) / 256)), dim3(256), 0, h->stream, A,
                     (long long)Np, (long long)lda, Af, (long long)ldaf);
  GPK_LAUNCH_CHECK(h);
  return GPK_OK;
}

extern "C" int gpk_wtw(gpk_handle h, const double* W, int64_t Np, int64_t ldw, double* Kinv, int64_t ldk) {
  if (!h) return GPK_BAD_ARG;
  GPK_REQUIRE(h, W && Kinv, "wtw: null pointer");
  GPK_REQUIRE(h, Np % NB == 0 && Np > 0 && ldw >= Np && ldk >= Np, "wtw: Np must be a multiple of 128");
  // Kinv (lower tiles) = W^T W: C[i][j] = sum_{k >= i} W[k][i] W[k][j]
  GemmArgs g = gemm_args(W, ldw, 1, W, ldw, 1, Kinv, ldk, (int)Np, (int)Np, (int)Np, 1.0, 0.0);
  g.lower_only = 1;
  g.kb_row = NB;
  g.k_super = 1;           // W (from gpk_trtri) is zero right of the diagonal for GPK_ZERO_BAND_TILES - 1 tiles
  return gpk_gemm(h, GPK_F64, g);
}

extern "C" int gpk_potri(gpk_handle h, const double* L, int64_t Np, int64_t ldl, const double* winv, double* Kinv,
                         int64_t ldk, double* work) {
  if (!h) return GPK_BAD_ARG;
  GPK_REQUIRE(h, L && winv && Kinv && work, "potri: null pointer");
  GPK_REQUIRE(h, Np % NB == 0 && Np > 0 && ldl >= Np && ldk >= Np, "potri: Np must be a multiple of 128");
  // work holds W = L^-1 (Np x Np: lower and diagonal tiles plus the band of zeros right of the diagonal that the
  // lockstep launches of gpk_trtri / gpk_wtw read - gpk_trtri writes both, `work` may hold anything on entry);
  // Kinv doubles as the recursion scratch ((Np/2 + 128)^2 <= Np^2 doubles for Np >= 256; unused at Np = 128)
  // before it is written
  double* W = work;
  GPK_TRY(gpk_trtri(h, L, Np, ldl, winv, W, Np, Kinv));
  return gpk_wtw(h, W, Np, Np, Kinv, ldk);
}

extern "C" int gpk_potrs_inv(gpk_handle h, const double* W, int64_t Np, int64_t ldw, const double* Y, int64_t N,
                             int P, double* alpha) {
  if (!h) return GPK_BAD_ARG;
  GPK_REQUIRE(h, W && Y && alpha, "potrs_inv: null pointer");
  GPK_REQUIRE(h, Np % NB == 0 && N >= 1 && N <= Np && ldw >= Np, "potrs_inv: bad sizes");
  GPK_REQUIRE(h, P >= 1 && P <= GPK_MAX_P, "potrs_inv: P must be in [1, 16]");
  const int nb = h->batch;
  // large models with a handful of targets: two streaming passes over W (HBM-bound) instead of two tile GEMMs on a
  // 128-column panel; small models keep the two GEMM launches (latency-bound there)
  if (nb == 1 && P <= 6 && Np >= h->k3_stream_min_np && ldw % 2 == 0 && ((uintptr_t)W % 16) == 0) {
    switch (P) {
      case 1: return k3_stream<1>(h, W, Np, ldw, Y, N, alpha);
      case 2: return k3_stream<2>(h, W, Np, ldw, Y, N, alpha);
      case 3: return k3_stream<3>(h, W, Np, ldw, Y, N, alpha);
      case 4: return k3_stream<4>(h, W, Np, ldw, Y, N, alpha);
      case 5: return k3_stream<5>(h, W, Np, ldw, Y, N, alpha);
      default: return k3_stream<6>(h, W, Np, ldw, Y, N, alpha);
    }
  }
  const long long panel = Np * NB;                         // doubles per problem and panel
  void* ws = nullptr;
  GPK_TRY(gpk_scratch(h, (size_t)2 * nb * panel * sizeof(double), &ws));
  double* Yp = (double*)ws;
  double* Z = Yp + (long long)nb * panel;
  const long long sY = gpk_bstride(h, Y) / (long long)sizeof(double), sAl = gpk_bstride(h, alpha) / (long long)sizeof(double);
  for (int b = 0; b < nb; ++b) {
    hipLaunchKernelGGL(pack_rhs_kernel, dim3((unsigned)((panel + 255) / 256)), dim3(256), 0, h->stream, Y + b * sY,
                       (long long)N, P, Yp + b * panel, (long long)Np);
    GPK_LAUNCH_CHECK(h);
  }
  // the two panels are per-problem scratch: register them for the duration of the two launches
  if (nb > 1) {
    h->bbufs.push_back({(const char*)Yp, panel * (long long)sizeof(double)});
    h->bbufs.push_back({(const char*)Z, panel * (long long)sizeof(double)});
  }
  // Z = W Yp   (W lower: k < row-tile end), then Yp = W^T Z   (k >= row-tile start): two launches
  GemmArgs g1 = gemm_args(W, ldw, 0, Yp, NB, 1, Z, NB, (int)Np, NB, (int)Np, 1.0, 0.0);
  g1.ke0 = NB; g1.ke_row = NB; g1.heavy_first = 1;
  int rc = gpk_gemm(h, GPK_F64, g1);
  if (rc == GPK_OK) {
    GemmArgs g2 = gemm_args(W, ldw, 1, Z, NB, 1, Yp, NB, (int)Np, NB, (int)Np, 1.0, 0.0);
    g2.kb_row = NB;
    rc = gpk_gemm(h, GPK_F64, g2);
  }
  if (nb > 1) { h->bbufs.pop_back(); h->bbufs.pop_back(); }
  GPK_TRY(rc);
  for (int b = 0; b < nb; ++b) {
    hipLaunchKernelGGL(unpack_rhs_kernel, dim3((unsigned)((N * P + 255) / 256)), dim3(256), 0, h->stream,
                       (const double*)(Yp + b * panel), (long long)N, P, alpha + b * sAl);
    GPK_LAUNCH_CHECK(h);
  }
  return GPK_OK;
}
