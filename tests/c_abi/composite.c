/* The composite C ABI (gpk_fit / gpk_predict / gpk_lml / gpk_export / gpk_import) from a plain C caller - no Python,
 * no torch - against known answers of scikit-learn and of the reference's ROS-package GP on the reference's flight
 * CSV.  The known answers come from tests/golden/known_answers.npz (KA2: sklearn, RBF(0.5) + White(0.1), alpha 1e-4,
 * normalize_y, D = 9, P = 3; KA5: the package GaussianProcess, ls = 1, sf2 = 1, noise 0.01, no normalisation); the
 * pytest wrapper (tests/test_gpu_c_abi.py) dumps them as one flat file of doubles, whose path is argv[1]:
 *   [N, D, P, M] X (N x D) Y (N x P) Xq (M x D)
 *   ka2: mean (M x P) std (M x P) lml theta (2) grad (2) alpha (N x P)      ka5: mean (M x P) var (M x P) lml
 *   (the last value in scikit-learn's form - log det once per output - from the repo's oracle: the package GP's own
 *   LML counts it once in total)
 * Bars: fp64 1e-8 relative (BASELINE.json north_star), fp32 serving 1e-4 (mean) / 1e-3 (std).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gpk.h"

#define CHECK_GPK(x) do { int r_ = (x); if (r_ != GPK_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, r_, gpk_last_error(h)); return 3; } } while (0)
#define EXPECT(cond, ...) do { if (!(cond)) { fprintf(stderr, "FAILED %s: ", #cond); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); return 4; } } while (0)

static double relerr(const double* a, const double* b, long n) {
  double num = 0.0, den = 1e-300;
  for (long i = 0; i < n; ++i) { if (fabs(a[i] - b[i]) > num) num = fabs(a[i] - b[i]); if (fabs(b[i]) > den) den = fabs(b[i]); }
  return num / den;
}

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s <known-answer file>\n", argv[0]); return 1; }
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 1; }
  fseek(f, 0, SEEK_END);
  const long bytes = ftell(f);
  fseek(f, 0, SEEK_SET);
  double* buf = (double*)malloc(bytes);
  if (fread(buf, 1, bytes, f) != (size_t)bytes) { fprintf(stderr, "short read\n"); return 1; }
  fclose(f);
  const long N = (long)buf[0], M = (long)buf[3];
  const int D = (int)buf[1], P = (int)buf[2];
  const double* X = buf + 4;
  const double* Y = X + N * D;
  const double* Xq = Y + N * P;
  const double* ka2_mean = Xq + M * D;
  const double* ka2_std = ka2_mean + M * P;
  const double ka2_lml = ka2_std[M * P];
  const double* ka2_theta = ka2_std + M * P + 1;
  const double* ka2_grad = ka2_theta + 2;
  const double* ka2_alpha = ka2_grad + 2;
  const double* ka5_mean = ka2_alpha + N * P;
  const double* ka5_var = ka5_mean + M * P;
  const double ka5_lml = ka5_var[M * P];
  EXPECT((ka5_var + M * P + 1 - buf) * (long)sizeof(double) == bytes, "file layout: %ld bytes", bytes);

  gpk_handle h = NULL;
  if (gpk_create(&h, 0) != GPK_OK) { fprintf(stderr, "gpk_create failed\n"); return 1; }
  CHECK_GPK(gpk_set_stream(h, GPK_OWN_STREAM));

  /* ---- scikit-learn surface (KA2) ------------------------------------------------------------------------- */
  const double ls = exp(ka2_theta[0]), noise = exp(ka2_theta[1]);
  EXPECT(fabs(ls - 0.5) < 1e-12 && fabs(noise - 0.1) < 1e-12, "theta %g %g", ls, noise);
  CHECK_GPK(gpk_fit(h, X, N, D, Y, P, &ls, 1, 1.0, noise, 1e-4, 1));
  double* mean = (double*)malloc(sizeof(double) * 8 * M * P);
  double* var = (double*)malloc(sizeof(double) * 8 * M * P);
  CHECK_GPK(gpk_predict(h, Xq, M, mean, var, GPK_F64, 1));               /* M = 64: the one-call serving path */
  for (long i = 0; i < M * P; ++i) var[i] = sqrt(var[i]);
  double e_mean = relerr(mean, ka2_mean, M * P), e_std = relerr(var, ka2_std, M * P);
  EXPECT(e_mean < 1e-8 && e_std < 1e-8, "KA2 fp64 (serving path): mean %.2e std %.2e", e_mean, e_std);
  /* the same queries five times over: the general panel path (320 rows) */
  double* Xq5 = (double*)malloc(sizeof(double) * 5 * M * D);
  for (int r = 0; r < 5; ++r) memcpy(Xq5 + r * M * D, Xq, sizeof(double) * M * D);
  CHECK_GPK(gpk_predict(h, Xq5, 5 * M, mean, var, GPK_F64, 1));
  for (int r = 0; r < 5; ++r) {
    for (long i = 0; i < M * P; ++i) var[r * M * P + i] = sqrt(var[r * M * P + i]);
    e_mean = relerr(mean + r * M * P, ka2_mean, M * P); e_std = relerr(var + r * M * P, ka2_std, M * P);
    EXPECT(e_mean < 1e-8 && e_std < 1e-8, "KA2 fp64 (panel path, copy %d): mean %.2e std %.2e", r, e_mean, e_std);
  }
  CHECK_GPK(gpk_predict(h, Xq5, 5 * M, mean, NULL, GPK_F64, 1));          /* means only */
  EXPECT(relerr(mean + 2 * M * P, ka2_mean, M * P) < 1e-8, "means-only call");
  /* fp32 serving */
  float* q32 = (float*)malloc(sizeof(float) * 5 * M * D);
  float* m32 = (float*)malloc(sizeof(float) * 5 * M * P);
  float* v32 = (float*)malloc(sizeof(float) * 5 * M * P);
  for (long i = 0; i < 5 * M * D; ++i) q32[i] = (float)Xq5[i];
  CHECK_GPK(gpk_predict(h, q32, 5 * M, m32, v32, GPK_F32, 1));
  double* ref32m = (double*)malloc(sizeof(double) * 5 * M * P);           /* fp64 answers at the fp32-rounded queries */
  double* ref32v = (double*)malloc(sizeof(double) * 5 * M * P);
  for (long i = 0; i < 5 * M * D; ++i) Xq5[i] = (double)q32[i];
  CHECK_GPK(gpk_predict(h, Xq5, 5 * M, ref32m, ref32v, GPK_F64, 1));
  double em = 0.0, es = 0.0, mx = 0.0;
  for (long i = 0; i < 5 * M * P; ++i) {
    if (fabs(ref32m[i]) > mx) mx = fabs(ref32m[i]);
    if (fabs((double)m32[i] - ref32m[i]) > em) em = fabs((double)m32[i] - ref32m[i]);
    const double s64 = sqrt(ref32v[i]), s32 = sqrt((double)v32[i]);
    if (fabs(s32 - s64) / s64 > es) es = fabs(s32 - s64) / s64;
  }
  EXPECT(em / mx < 1e-4 && es < 1e-3, "fp32 serving: mean %.2e std %.2e", em / mx, es);
  /* log-marginal likelihood: the fitted value, the value and gradient at theta, a not-PD trial point */
  double lml = 0.0, lml2 = 0.0, grad[2] = {0.0, 0.0};
  CHECK_GPK(gpk_lml(h, NULL, 0, &lml, NULL));
  EXPECT(fabs(lml - ka2_lml) < 1e-10 * fabs(ka2_lml), "fitted LML %.12g vs %.12g", lml, ka2_lml);
  CHECK_GPK(gpk_lml(h, ka2_theta, 2, &lml2, grad));
  EXPECT(fabs(lml2 - ka2_lml) < 1e-10 * fabs(ka2_lml), "LML(theta) %.12g", lml2);
  EXPECT(relerr(grad, ka2_grad, 2) < 1e-8, "gradient %.10g %.10g vs %.10g %.10g", grad[0], grad[1], ka2_grad[0], ka2_grad[1]);
  double ard_theta[17], ard_grad[17], s = 0.0;
  for (int d = 0; d < D; ++d) ard_theta[d] = ka2_theta[0];
  ard_theta[D] = ka2_theta[1];
  CHECK_GPK(gpk_lml(h, ard_theta, D + 1, &lml2, ard_grad));              /* ARD layout: per-feature gradients sum to the isotropic one */
  for (int d = 0; d < D; ++d) s += ard_grad[d];
  EXPECT(fabs(s - ka2_grad[0]) < 1e-8 * fabs(ka2_grad[0]) && fabs(ard_grad[D] - ka2_grad[1]) < 1e-8 * fabs(ka2_grad[1]), "ARD gradient sum %.10g", s);
  CHECK_GPK(gpk_predict(h, Xq, M, mean, NULL, GPK_F64, 1));               /* the fitted factor survived the trial evaluations */
  EXPECT(relerr(mean, ka2_mean, M * P) < 1e-8, "fitted model after gpk_lml");
  /* export -> import into a second handle -> same predictions, alpha as scikit-learn's */
  double* L = (double*)malloc(sizeof(double) * N * N);
  double* alpha = (double*)malloc(sizeof(double) * N * P);
  double ym[16], ys[16], elml = 0.0;
  int64_t eN = 0; int eD = 0, eP = 0;
  CHECK_GPK(gpk_export(h, &eN, &eD, &eP, L, alpha, ym, ys, &elml));
  EXPECT(eN == N && eD == D && eP == P && elml == lml, "export header");
  EXPECT(relerr(alpha, ka2_alpha, N * P) < 1e-8, "alpha %.2e", relerr(alpha, ka2_alpha, N * P));
  EXPECT(L[1] == 0.0 && L[N + 1] > 0.0, "L is lower triangular");
  gpk_handle h2 = NULL;
  if (gpk_create(&h2, 0) != GPK_OK) return 1;
  { gpk_handle h = h2;
    CHECK_GPK(gpk_set_stream(h, GPK_OWN_STREAM));
    CHECK_GPK(gpk_import(h, X, N, D, L, alpha, P, &ls, 1, 1.0, noise, ym, ys));
    CHECK_GPK(gpk_predict(h, Xq, M, mean, var, GPK_F64, 1));
    for (long i = 0; i < M * P; ++i) var[i] = sqrt(var[i]);
    EXPECT(relerr(mean, ka2_mean, M * P) < 1e-8 && relerr(var, ka2_std, M * P) < 1e-8, "imported model");
    EXPECT(gpk_lml(h, ka2_theta, 2, &lml2, NULL) == GPK_BAD_ARG, "lml(theta) on an imported model must be refused");
    EXPECT(gpk_predict(h, Xq, M, mean, var, 7, 1) == GPK_BAD_ARG, "bad dtype");
  }
  gpk_destroy(h2);

  /* ---- ROS-package surface (KA5): no normalisation, k** = sf2, variance floored at 1e-10, one value per query */
  const double one = 1.0;
  CHECK_GPK(gpk_fit(h, X, N, D, Y, P, &one, 1, 1.0, 0.01, 0.0, 0));
  CHECK_GPK(gpk_predict(h, Xq, M, mean, var, GPK_F64, 0));
  EXPECT(relerr(mean, ka5_mean, M * P) < 1e-8 && relerr(var, ka5_var, M * P) < 1e-8, "KA5: mean %.2e var %.2e",
         relerr(mean, ka5_mean, M * P), relerr(var, ka5_var, M * P));
  CHECK_GPK(gpk_lml(h, NULL, 0, &lml, NULL));
  EXPECT(fabs(lml - ka5_lml) < 1e-10 * fabs(ka5_lml), "KA5 LML %.12g vs %.12g", lml, ka5_lml);

  /* ---- failure conventions ---------------------------------------------------------------------------------- */
  double* Xd = (double*)malloc(sizeof(double) * 40 * D);
  double* Yd = (double*)malloc(sizeof(double) * 40 * P);
  for (long i = 0; i < 20 * D; ++i) Xd[i] = Xd[20 * D + i] = X[i];        /* exact duplicates, no noise: singular */
  for (long i = 0; i < 40 * P; ++i) Yd[i] = Y[i];
  EXPECT(gpk_fit(h, Xd, 40, D, Yd, P, &one, 1, 1.0, 0.0, 0.0, 1) == GPK_NOT_PD, "duplicates without noise must be GPK_NOT_PD");
  EXPECT(gpk_predict(h, Xq, M, mean, var, GPK_F64, 1) == GPK_BAD_ARG, "predict after a failed fit must be refused");
  CHECK_GPK(gpk_fit(h, Xd, 40, D, Yd, P, &one, 1, 1.0, 0.05, 0.0, 1));   /* with noise the same data fit */
  double th[2] = {0.0, -80.0};                                            /* noise e^-80: the trial matrix is singular */
  CHECK_GPK(gpk_lml(h, th, 2, &lml, grad));
  EXPECT(isinf(lml) && lml < 0 && grad[0] == 0.0 && grad[1] == 0.0, "not-PD trial point: -inf / zero gradient, got %g", lml);
  Xd[3] = NAN;
  EXPECT(gpk_fit(h, Xd, 40, D, Yd, P, &one, 1, 1.0, 0.05, 0.0, 1) == GPK_BAD_ARG, "NaN inputs");
  CHECK_GPK(gpk_model_release(h));
  gpk_destroy(h);
  printf("C ABI composite: OK\n");
  return 0;
}
