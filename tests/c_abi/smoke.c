/* C-ABI smoke test: a plain C caller (no Python, no torch) drives libgpk.so through include/gpk.h with
 * hipMalloc'd buffers: Gram build (K1), Cholesky (K2), alpha solve (K3), posterior mean (K4) and variance (K5)
 * on a small synthetic problem, checked through properties that need no reference implementation:
 *   (K + s I) alpha = y   =>   mean(x_i) = y_i - s alpha_i   at every training point,
 *   0 <= var(x_i) < s (1 + tolerance) + ... i.e. the posterior variance at a training point is below the prior,
 *   a non-positive-definite matrix is reported as GPK_NOT_PD.
 * Build: gcc -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude smoke.c -L<pkg> -lgpk -L/opt/rocm/lib -lamdhip64 -lm
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "gpk.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_GPK(x) do { int r_ = (x); if (r_ != GPK_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, r_, gpk_last_error(h)); return 3; } } while (0)

int main(void) {
  const int64_t N = 300;
  const int D = 3, P = 2;
  const double ls[3] = {0.9, 1.1, 1.3}, sf2 = 1.2, s = 0.05;
  double* X = (double*)malloc(sizeof(double) * N * D);
  double* Y = (double*)malloc(sizeof(double) * N * P);
  unsigned long long st = 88172645463325252ull;
  for (int64_t i = 0; i < N * D; ++i) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; X[i] = (double)(st % 2000001ull) / 1e6 - 1.0; }
  for (int64_t i = 0; i < N; ++i) { Y[i * P] = sin(2.0 * X[i * D]) + X[i * D + 1]; Y[i * P + 1] = cos(X[i * D + 2]) * X[i * D]; }

  gpk_handle h = NULL;
  if (gpk_create(&h, 0) != GPK_OK) { fprintf(stderr, "gpk_create failed\n"); return 1; }
  CHECK_GPK(gpk_set_stream(h, GPK_OWN_STREAM));
  const int64_t Np = gpk_padded(N);
  double *dX, *dY, *dK, *dwinv, *dalpha, *dmean, *dvar, *dwork, *dW, *dtr;
  CHECK_HIP(hipMalloc((void**)&dX, sizeof(double) * N * D));
  CHECK_HIP(hipMalloc((void**)&dY, sizeof(double) * N * P));
  CHECK_HIP(hipMalloc((void**)&dK, sizeof(double) * Np * Np));
  CHECK_HIP(hipMalloc((void**)&dW, sizeof(double) * Np * Np));
  CHECK_HIP(hipMalloc((void**)&dtr, sizeof(double) * (Np / 2 + 128) * (Np / 2 + 128)));
  CHECK_HIP(hipMalloc((void**)&dwinv, sizeof(double) * Np * GPK_TILE));
  CHECK_HIP(hipMalloc((void**)&dalpha, sizeof(double) * N * P));
  CHECK_HIP(hipMalloc((void**)&dmean, sizeof(double) * N * P));
  CHECK_HIP(hipMalloc((void**)&dvar, sizeof(double) * Np));
  CHECK_HIP(hipMalloc((void**)&dwork, sizeof(double) * Np * Np));
  CHECK_HIP(hipMemcpy(dX, X, sizeof(double) * N * D, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(dY, Y, sizeof(double) * N * P, hipMemcpyHostToDevice));

  int info = -1;
  CHECK_GPK(gpk_gram(h, GPK_F64, dX, N, D, ls, sf2, s, dK, Np));
  CHECK_GPK(gpk_potrf(h, dK, Np, Np, dwinv, &info));
  if (info != 0) { fprintf(stderr, "potrf info %d\n", info); return 4; }
  CHECK_GPK(gpk_potrs(h, dK, Np, Np, dwinv, dY, N, P, dalpha));
  const double zero[2] = {0.0, 0.0}, one[2] = {1.0, 1.0};
  CHECK_GPK(gpk_predict_mean(h, GPK_F64, dX, dalpha, N, D, P, ls, sf2, zero, one, dX, N, dmean));
  CHECK_GPK(gpk_trtri(h, dK, Np, Np, dwinv, dW, Np, dtr));
  CHECK_GPK(gpk_predict_var_inv(h, GPK_F64, dX, N, D, ls, sf2, dW, Np, Np, dX, N, sf2 + s, 0.0, dwork, dvar));
  CHECK_GPK(gpk_synchronize(h));

  double* alpha = (double*)malloc(sizeof(double) * N * P);
  double* mean = (double*)malloc(sizeof(double) * N * P);
  double* var = (double*)malloc(sizeof(double) * N);
  CHECK_HIP(hipMemcpy(alpha, dalpha, sizeof(double) * N * P, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(mean, dmean, sizeof(double) * N * P, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(var, dvar, sizeof(double) * N, hipMemcpyDeviceToHost));
  double emax = 0.0, vmin = 1e300, vmax = -1e300;
  for (int64_t i = 0; i < N * P; ++i) { const double e = fabs(mean[i] - (Y[i] - s * alpha[i])); if (e > emax) emax = e; }
  for (int64_t i = 0; i < N; ++i) { if (var[i] < vmin) vmin = var[i]; if (var[i] > vmax) vmax = var[i]; }
  printf("max |mean - (y - s alpha)| = %.3e   var at training points in [%.3e, %.3e] (prior %.3f)\n", emax, vmin, vmax, sf2 + s);
  if (!(emax < 1e-10)) return 5;
  if (!(vmin >= 0.0 && vmax < 2.0 * s && vmin > 0.5 * s)) return 6;

  /* the one-call serving entry points: host queries in, host results out (5 rows -> the two-launch small-batch
   * kernels), one model with P outputs and two single-output models in one call */
  {
    const int64_t M = 5;
    double hm[5 * 2], hv[5], mm[2 * 5], mv[2 * 5];
    CHECK_GPK(gpk_predict_host(h, dX, dalpha, N, D, P, ls, sf2, zero, one, dW, Np, Np, sf2 + s, 0.0, X, M, hm, hv));
    double e1 = 0.0, e2 = 0.0;
    for (int64_t i = 0; i < M * P; ++i) e1 = fmax(e1, fabs(hm[i] - mean[i]));
    for (int64_t i = 0; i < M; ++i) e2 = fmax(e2, fabs(hv[i] - var[i]));
    printf("gpk_predict_host vs the device chain: mean %.2e var %.2e\n", e1, e2);
    if (!(e1 < 1e-11 && e2 < 1e-11)) return 8;
    double *da0, *da1, *dy0, *dy1;
    double* ycol = (double*)malloc(sizeof(double) * N);
    CHECK_HIP(hipMalloc((void**)&da0, sizeof(double) * N)); CHECK_HIP(hipMalloc((void**)&da1, sizeof(double) * N));
    CHECK_HIP(hipMalloc((void**)&dy0, sizeof(double) * N)); CHECK_HIP(hipMalloc((void**)&dy1, sizeof(double) * N));
    for (int64_t i = 0; i < N; ++i) ycol[i] = Y[i * P];
    CHECK_HIP(hipMemcpy(dy0, ycol, sizeof(double) * N, hipMemcpyHostToDevice));
    for (int64_t i = 0; i < N; ++i) ycol[i] = Y[i * P + 1];
    CHECK_HIP(hipMemcpy(dy1, ycol, sizeof(double) * N, hipMemcpyHostToDevice));
    CHECK_GPK(gpk_potrs(h, dK, Np, Np, dwinv, dy0, N, 1, da0));
    CHECK_GPK(gpk_potrs(h, dK, Np, Np, dwinv, dy1, N, 1, da1));
    const double* Xs[2] = {dX, dX};
    const double* As[2] = {da0, da1};
    const double* Ws[2] = {dW, dW};
    const double ls2[6] = {0.9, 1.1, 1.3, 0.9, 1.1, 1.3}, sf22[2] = {sf2, sf2}, kss2[2] = {sf2 + s, sf2 + s};
    CHECK_GPK(gpk_predict_host_multi(h, 2, Xs, As, N, D, ls2, sf22, zero, one, Ws, Np, Np, kss2, 0.0, X, M, mm, mv));
    double e3 = 0.0, e4 = 0.0;
    for (int b = 0; b < 2; ++b)
      for (int64_t i = 0; i < M; ++i) {
        e3 = fmax(e3, fabs(mm[b * M + i] - mean[i * P + b]));
        e4 = fmax(e4, fabs(mv[b * M + i] - var[i]));
      }
    printf("gpk_predict_host_multi (2 models) vs the device chain: mean %.2e var %.2e\n", e3, e4);
    if (!(e3 < 1e-11 && e4 < 1e-11)) return 9;
    if (gpk_predict_host_multi(h, 2, Xs, As, N, D, ls2, sf22, zero, one, Ws, Np, Np, kss2, 0.0, X, 33, mm, mv) != GPK_BAD_ARG) return 10;
    if (gpk_predict_host_multi(h, 9, Xs, As, N, D, ls2, sf22, zero, one, Ws, Np, Np, kss2, 0.0, X, M, mm, mv) != GPK_BAD_ARG) return 11;
  }

  /* not positive definite: a Gram matrix with a negative "noise" large enough to break it */
  CHECK_GPK(gpk_gram(h, GPK_F64, dX, N, D, ls, sf2, -sf2 - 1.0, dK, Np));
  const int rc = gpk_potrf(h, dK, Np, Np, dwinv, &info);
  printf("indefinite matrix: rc = %d, info = %d (%s)\n", rc, info, gpk_last_error(h));
  if (rc != GPK_NOT_PD || info <= 0) return 7;

  gpk_destroy(h);
  printf("C ABI smoke: OK\n");
  return 0;
}
