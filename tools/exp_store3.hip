// Experiment: write bandwidth of the SYMMETRIC Gram store pattern - every lower-triangle tile (ti >= tj) is written at
// [ti][tj] and, mirrored, at [tj][ti] - under different orders in which the tiles are dealt to workgroups.
//   hipcc --offload-arch=gfx950 -O3 tools/exp_store3.hip -o /tmp/exp_store3 && /tmp/exp_store3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double dv2 __attribute__((ext_vector_type(2)));

template <int TS>
__device__ __forceinline__ void put(double* K, long long N, long long ti, long long tj, double v0) {
  constexpr int LPR = TS / 2, RPP = 256 / LPR;
  const int lp = threadIdx.x % LPR, r0 = threadIdx.x / LPR;
  for (int r = r0; r < TS; r += RPP) {
    dv2 v = {v0 + r, v0 + lp};
    *(dv2*)(K + (ti * TS + r) * N + tj * TS + 2 * lp) = v;
  }
}
// ORDER 0: row-major over the lower triangle (ti outer);  1: column-major (tj outer);  2: diagonal by diagonal (ti - tj outer);
// 3: row-major, GS = 8 column tiles per workgroup (the strip kernel's pattern)
template <int TS, int ORDER>
__global__ __launch_bounds__(256) void sym_store(double* K, long long N, long long nt) {
  long long id = blockIdx.x, ti, tj;
  if (ORDER == 0 || ORDER == 3) {
    const long long per = ORDER == 3 ? 8 : 1;
    if (ORDER == 3) {
      // strips: tile row r has r / 8 + 1 strips
      long long q = (long long)((__builtin_sqrt(1.0 + 8.0 * (double)id / 8) - 1.0) * 0.5);
      while (4 * (q + 1) * (q + 2) <= id) ++q;
      while (4 * q * (q + 1) > id) --q;
      const long long rem = id - 4 * q * (q + 1);
      ti = 8 * q + rem / (q + 1);
      const long long g = rem % (q + 1);
      for (long long t = g * 8; t <= ti && t < g * 8 + 8; ++t) { put<TS>(K, N, ti, t, 1.0); if (t != ti) put<TS>(K, N, t, ti, 2.0); }
      return;
    }
    ti = (long long)((__builtin_sqrt(8.0 * (double)id + 1.0) - 1.0) * 0.5);
    while ((ti + 1) * (ti + 2) / 2 <= id) ++ti;
    while (ti * (ti + 1) / 2 > id) --ti;
    tj = id - ti * (ti + 1) / 2;
    (void)per;
  } else if (ORDER == 1) {
    // column tj has nt - tj tiles; columns in order
    long long c = 0, rem = id;
    // closed form: tiles before column c = c nt - c (c - 1) / 2
    c = (long long)(((2.0 * nt + 1.0) - __builtin_sqrt((2.0 * nt + 1.0) * (2.0 * nt + 1.0) - 8.0 * (double)id)) * 0.5);
    while (c * nt - c * (c - 1) / 2 > id) --c;
    while ((c + 1) * nt - (c + 1) * c / 2 <= id) ++c;
    rem = id - (c * nt - c * (c - 1) / 2);
    tj = c; ti = c + rem;
  } else {
    // diagonal d = ti - tj has nt - d tiles; diagonals in order, tiles along a diagonal consecutive
    long long d = (long long)(((2.0 * nt + 1.0) - __builtin_sqrt((2.0 * nt + 1.0) * (2.0 * nt + 1.0) - 8.0 * (double)id)) * 0.5);
    while (d * nt - d * (d - 1) / 2 > id) --d;
    while ((d + 1) * nt - (d + 1) * d / 2 <= id) ++d;
    const long long rem = id - (d * nt - d * (d - 1) / 2);
    tj = rem; ti = rem + d;
  }
  put<TS>(K, N, ti, tj, 1.0);
  if (ti != tj) put<TS>(K, N, tj, ti, 2.0);
}
template <typename F> float timeit(F f, int it = 5) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  f(); (void)hipDeviceSynchronize();
  float best = 1e30f;
  for (int i = 0; i < it; ++i) { (void)hipEventRecord(a); f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b); float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
  return best;
}
int main() {
  const long long N = 65536; double* K; (void)hipMalloc(&K, N * N * 8);
  const double gb = N * N * 8 / 1e9;
  float ms;
#define RUN(TS, ORD, name) { const long long nt = N / TS; const long long nb = ORD == 3 ? ({ long long s = 0; for (long long r = 0; r < nt; ++r) s += r / 8 + 1; s; }) : nt * (nt + 1) / 2; \
  ms = timeit([&] { hipLaunchKernelGGL((sym_store<TS, ORD>), dim3((unsigned)nb), dim3(256), 0, 0, K, N, nt); }); \
  printf("symmetric %3d x %3d tiles, %-28s: %.3f ms  %.0f GB/s\n", TS, TS, name, ms, gb / ms * 1e3); }
  RUN(64, 0, "row-major") RUN(64, 3, "row-major strips of 8") RUN(64, 1, "column-major") RUN(64, 2, "diagonal by diagonal")
  RUN(128, 0, "row-major") RUN(128, 1, "column-major") RUN(128, 2, "diagonal by diagonal")
  RUN(32, 2, "diagonal by diagonal")
  return 0;
}
