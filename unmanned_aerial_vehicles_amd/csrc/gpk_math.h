// Device math shared by the RBF kernels (gfx950).
#pragma once
#include <hip/hip_runtime.h>

// exp(x) for x <= 0 in fp64: k = rint(x log2e), r = x - k ln2 (hi/lo split, FMA), Taylor degree 13 on
// |r| <= ln2/2 (truncation 4e-18 relative), scaled by 2^k with v_ldexp_f64 (gradual underflow to 0).
__device__ __forceinline__ double gpk_exp_neg(double x) {
  x = fmax(x, -800.0);
  const double k = __builtin_rint(x * 1.4426950408889634);
  double r = __builtin_fma(k, -6.93147180559945286227e-01, x);
  r = __builtin_fma(k, -2.31904681384629955842e-17, r);
  double p = 1.6059043836821613e-10;                   // 1/13!
  p = __builtin_fma(p, r, 2.08767569878680990e-09);    // 1/12!
  p = __builtin_fma(p, r, 2.50521083854417188e-08);    // 1/11!
  p = __builtin_fma(p, r, 2.75573192239858907e-07);    // 1/10!
  p = __builtin_fma(p, r, 2.75573192239858907e-06);    // 1/9!
  p = __builtin_fma(p, r, 2.48015873015873016e-05);    // 1/8!
  p = __builtin_fma(p, r, 1.98412698412698413e-04);    // 1/7!
  p = __builtin_fma(p, r, 1.38888888888888889e-03);    // 1/6!
  p = __builtin_fma(p, r, 8.33333333333333333e-03);    // 1/5!
  p = __builtin_fma(p, r, 4.16666666666666667e-02);    // 1/4!
  p = __builtin_fma(p, r, 1.66666666666666667e-01);    // 1/3!
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  return __builtin_ldexp(p, (int)k);
}
