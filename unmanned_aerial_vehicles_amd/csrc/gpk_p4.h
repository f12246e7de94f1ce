// Cholesky factor AND inverse of one 16 x 16 diagonal block by ONE wave, on the accumulator itself (gfx950).
//
// The block arrives as the one-launch Cholesky keeps it (gpk_ptile.hip): U[t], lane (n = lane & 15, q = lane >> 4) = entry
// (row n, column 4 t + q) - the transposed block in the f64 C/D map, both triangles valid.  Register t is a 16 x 4 PANEL
// (columns 4 t .. 4 t + 3) and, as it stands, both the A operand (lane (m, k): element [m][k]) and the B operand (lane (n, k):
// element [k][n] of the transpose) of v_mfma_f64_16x16x4_f64.  Right-looking by panels, identity rows X riding along (they end
// as L^-T, i.e. X[t], lane (n, q) = W[4 t + q][n]):
//   1. the 4 x 4 diagonal sub-block (ten v_readlane pairs: wave-uniform scalars) is factored and inverted in closed form -
//      four reciprocal square roots (v_rsq_f64 + one third-order correction), two dozen multiply-adds;
//   2. panel <- panel W_pp^T:  one MFMA (A = W_pp in rows 4 p .. 4 p + 3, B = the panel register);
//   3. the columns right of the panel:  block -= panel panel^T:  one MFMA (A = B = the panel register, A zeroed for the
//      columns already final) - for U and for X each.
// Four panel steps of ~45 dependent scalar operations and two dependent MFMAs instead of the 16 column steps and ~750
// dependent vector instructions (272 of them 64-bit DPP broadcasts) of the row-per-lane sweep it replaces (gpk_p2.h).
// A non-positive pivot is replaced by 1 (the factor stays finite) and its 1-based column returned.
#pragma once
#include <hip/hip_runtime.h>

typedef double gpk_p4_d4 __attribute__((ext_vector_type(4)));
template <int V> struct GpkP4C { static constexpr int value = V; };
template <int I, int N, class F>
__device__ __forceinline__ void gpk_p4_for(F&& f) {
  if constexpr (I < N) {
    f(GpkP4C<I>{});
    gpk_p4_for<I + 1, N>(f);
  }
}
// the value lane LANE holds, in every lane (through the scalar registers)
template <int LANE>
__device__ __forceinline__ double gpk_p4_lane(double v) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), LANE), hi = __builtin_amdgcn_readlane(__double2hiint(v), LANE);
  return __hiloint2double(hi, lo);
}
// r = 1 / sqrt(t) for t > 0: v_rsq_f64 (good to 2^-24, measured) and ONE third-order correction
// y (1 + e / 2 + 3 e^2 / 8), e = 1 - t y^2: five dependent operations, error e^3 ~ 2^-72
__device__ __forceinline__ double gpk_p4_rsqrt(double t) {
  const double y = __builtin_amdgcn_rsq(t);
  const double e = __builtin_fma(-(t * y), y, 1.0);
  const double c = e * __builtin_fma(0.375, e, 0.5);
  return __builtin_fma(y, c, y);
}

__device__ __forceinline__ int gpk_p4_factor(gpk_p4_d4& U, gpk_p4_d4& X, int lane) {
  const int n = lane & 15, q = lane >> 4, a = n & 3;
  // 0 / 1 masks of this lane's place in a 4 x 4 sub-block (row a = n & 3, column q): the operand entries below are picked
  // by multiply-adds with them (exact: one term is the entry, the others are +-0) - as nested selects the compiler turns
  // them into branches
  double m[10];                                // (a, q) = (0,0) (1,0) (1,1) (2,0) (2,1) (2,2) (3,0) (3,1) (3,2) (3,3)
  {
    int k = 0;
#pragma unroll
    for (int aa = 0; aa < 4; ++aa)
#pragma unroll
      for (int qq = 0; qq <= aa; ++qq) {
        double v = (a == aa && q == qq) ? 1.0 : 0.0;
        asm volatile("" : "+v"(v));
        m[k++] = v;
      }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    double e = (n == 4 * t + q) ? 1.0 : 0.0;
    asm volatile("" : "+v"(e));                  // (opaque: keeps the identity from being folded into the first products)
    X[t] = e;
  }
  unsigned fail = 0;                             // bit c: pivot c was not positive
  gpk_p4_for<0, 4>([&](auto pc) {
    constexpr int P = decltype(pc)::value;
    constexpr int B = 4 * P;
    // ---- 1. the diagonal sub-block: entry (a, c) is in lane (n = B + a, q = c) of register P
    const double up = U[P];
    const double a00 = gpk_p4_lane<B>(up);
    const double a10 = gpk_p4_lane<B + 1>(up), a11 = gpk_p4_lane<16 + B + 1>(up);
    const double a20 = gpk_p4_lane<B + 2>(up), a21 = gpk_p4_lane<16 + B + 2>(up), a22 = gpk_p4_lane<32 + B + 2>(up);
    const double a30 = gpk_p4_lane<B + 3>(up), a31 = gpk_p4_lane<16 + B + 3>(up), a32 = gpk_p4_lane<32 + B + 3>(up),
                 a33 = gpk_p4_lane<48 + B + 3>(up);
    // (masks of this panel, formed while the pivots are on their way: rows of the panel, rows below it)
    const double inp = (n >> 2) == P ? 1.0 : 0.0, below = n >= B + 4 ? -1.0 : 0.0;
    auto pivot = [&](double t, int col) -> double {          // 1 / sqrt(pivot); a non-positive pivot counts as 1
      const bool ok = t > 0.0;                               // (beside the chain: it only selects the result)
      fail |= ok ? 0u : (1u << col);
      const double r = gpk_p4_rsqrt(t);
      return ok ? r : 1.0;
    };
    const double r0 = pivot(a00, B);
    const double l10 = a10 * r0, l20 = a20 * r0, l30 = a30 * r0;
    const double r1 = pivot(__builtin_fma(-l10, l10, a11), B + 1);
    const double l21 = __builtin_fma(-l20, l10, a21) * r1, l31 = __builtin_fma(-l30, l10, a31) * r1;
    const double r2 = pivot(__builtin_fma(-l21, l21, __builtin_fma(-l20, l20, a22)), B + 2);
    const double l32 = __builtin_fma(-l31, l21, __builtin_fma(-l30, l20, a32)) * r2;
    // W_pp = L_pp^-1 (lower, closed form); the entries of its last row are  r3 x (something known before r3)
    const double w10 = -l10 * r0 * r1, w21 = -l21 * r1 * r2;
    const double w20 = -__builtin_fma(l20, r0, l21 * w10) * r2;
    const double g32 = -l32 * r2, g31 = -__builtin_fma(l31, r1, l32 * w21), g30 = -__builtin_fma(l30, r0, __builtin_fma(l31, w10, l32 * w20));
    // ---- 2. panel <- panel W_pp^T: this lane's entry W_pp[a][q] of the A operand (rows B .. B + 3, zero elsewhere):
    //      early + r3 x late, both summed before the last pivot is there
    const double early = inp * __builtin_fma(m[0], r0, __builtin_fma(m[1], w10, __builtin_fma(m[2], r1, __builtin_fma(m[3], w20,
                               __builtin_fma(m[4], w21, m[5] * r2)))));
    const double late = inp * __builtin_fma(m[6], g30, __builtin_fma(m[7], g31, __builtin_fma(m[8], g32, m[9])));
    const double r3 = pivot(__builtin_fma(-l32, l32, __builtin_fma(-l31, l31, __builtin_fma(-l30, l30, a33))), B + 3);
    const double wop = __builtin_fma(r3, late, early);
    const gpk_p4_d4 zero = {0.0, 0.0, 0.0, 0.0};
    const gpk_p4_d4 su = __builtin_amdgcn_mfma_f64_16x16x4f64(wop, U[P], zero, 0, 0, 0);
    const gpk_p4_d4 sx = __builtin_amdgcn_mfma_f64_16x16x4f64(wop, X[P], zero, 0, 0, 0);
    U[P] = su[P];
    X[P] = sx[P];
    // ---- 3. the columns right of the panel
    if constexpr (P < 3) {
      const double aop = below * U[P];
      U = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, U[P], U, 0, 0, 0);
      X = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, X[P], X, 0, 0, 0);
    }
  });
  return fail ? __builtin_ctz(fail) + 1 : 0;
}
