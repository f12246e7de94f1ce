// Cycle count of the 16 x 16 factor + inverse routine of the one-launch Cholesky (gpk_p4.h), one wave, straight-line, and its
// result against a host factorisation.  (The row-per-lane DPP sweep it replaced - gpk_p2.h, in the history up to round 4 -
// measured 5 600-5 700 cycles in this harness; the panel routine 3 500.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I unmanned_aerial_vehicles_amd/csrc tools/exp_p4.hip -o /tmp/exp_p4 && /tmp/exp_p4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include "gpk_p4.h"
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int BS = 17, BLK = 16 * BS;

__global__ void k(const double* A, double* out, long long* cyc) {
  __shared__ double lds[4 * BLK];
  const int lane = threadIdx.x, n = lane & 15, q = lane >> 4;
  d4 U0;
  for (int t = 0; t < 4; ++t) U0[t] = A[n * 16 + 4 * t + q];
  // ---- panel routine
  d4 U = U0, X;
  long long t0 = clock64();
  int bad = 0;
  for (int r = 0; r < 64; ++r) {
    U = U0;
    asm volatile("" : "+v"(U));
    bad += gpk_p4_factor(U, X, lane);
    asm volatile("" :: "v"(U), "v"(X));
  }
  long long t1 = clock64();
  if (lane == 0) cyc[0] = (t1 - t0) / 64;
  for (int t = 0; t < 4; ++t) { out[n * 16 + 4 * t + q] = U[t]; out[256 + n * 16 + 4 * t + q] = X[t]; }
  if (lane == 0) { cyc[1] = 0; cyc[2] = bad; }
}

int main() {
  double hA[256], L[256] = {0};
  for (int i = 0; i < 16; ++i) for (int j = 0; j <= i; ++j) L[i * 16 + j] = (i == j) ? 1.0 + 0.1 * i : 0.05 * std::sin(1.0 + i * 3 + j);
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 16; ++k) s += L[i * 16 + k] * L[j * 16 + k]; hA[i * 16 + j] = s; }
  double *dA, *dO; long long* dC;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dO, 1024 * 8); hipMalloc(&dC, 3 * 8);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dO, dC);
  double hO[1024]; long long hC[3];
  hipMemcpy(hO, dO, sizeof hO, hipMemcpyDeviceToHost); hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
  double e4 = 0, w = 0;
  for (int i = 0; i < 16; ++i) for (int j = 0; j <= i; ++j) e4 = std::fmax(e4, std::fabs(hO[i * 16 + j] - L[i * 16 + j]));
  // X[n][c] = W[c][n]: W L = I
  for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c) { double a = 0; for (int k = 0; k < 16; ++k) a += hO[256 + k * 16 + r] * L[k * 16 + c]; w = std::fmax(w, std::fabs(a - (r == c))); }
  printf("panel routine %lld cycles (shader clock); |L - L_host| %.1e  |W L - I| %.1e  bad %lld\n", hC[0], e4, w, hC[2]);
  return 0;
}
