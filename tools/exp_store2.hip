// Experiment: HBM write bandwidth of row-segment store patterns with and without a fp64 vector-ALU load per entry
// (models a Gram build that computes EVERY entry directly - no mirrored tile - and writes long row segments).
//   hipcc --offload-arch=gfx950 -O3 tools/exp_store2.hip -o /tmp/exp_store2 && /tmp/exp_store2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double dv2 __attribute__((ext_vector_type(2)));

// each workgroup writes TR rows x TC doubles; FL dependent fp64 FMAs per stored entry; tiles row-major or column-major
template <int TR, int TC, int FL, bool COLMAJOR>
__global__ __launch_bounds__(256) void tile_store(double* K, long long N, long long tpr, long long tpc, double seed) {
  const long long t = blockIdx.x;
  const long long ti = COLMAJOR ? t % tpc : t / tpr, tj = COLMAJOR ? t / tpc : t % tpr;
  constexpr int LPR = TC / 2 < 256 ? TC / 2 : 256;   // lanes per row pass (16 B each)
  constexpr int RPP = 256 / LPR;                      // rows per pass
  constexpr int CP = TC / 2 / LPR;                    // column passes per row
  const int lp = threadIdx.x % LPR, r0 = threadIdx.x / LPR;
  for (int r = r0; r < TR; r += RPP)
    for (int c = 0; c < CP; ++c) {
      double a = seed + r, b = seed * lp + c;
#pragma unroll
      for (int f = 0; f < FL; ++f) { a = __builtin_fma(a, 1.0000001, b); b = __builtin_fma(b, 0.9999999, a); }
      double* p = K + (ti * TR + r) * N + tj * TC + 2 * (lp + c * LPR);
      dv2 v = {a, b};
      *(dv2*)p = v;
    }
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
#define RUN(TR, TC, FL, CM) ms = timeit([&] { hipLaunchKernelGGL((tile_store<TR, TC, FL, CM>), dim3((unsigned)((N / TR) * (N / TC))), dim3(256), 0, 0, K, N, N / TC, N / TR, 1.0); }); \
  printf("tile %3d x %4d doubles (%5d B rows) fma/entry %2d %s: %.3f ms  %.0f GB/s\n", TR, TC, TC * 8, 2 * FL / 2, CM ? "col-major" : "row-major", ms, gb / ms * 1e3);
  RUN(8, 512, 0, false) RUN(4, 512, 0, false) RUN(16, 512, 0, false) RUN(32, 512, 0, false) RUN(8, 1024, 0, false) RUN(2, 2048, 0, false)
  RUN(8, 512, 0, true) RUN(16, 512, 0, true) RUN(64, 64, 0, false) RUN(64, 64, 0, true) RUN(128, 128, 0, true) RUN(64, 512, 0, false)
  RUN(8, 512, 8, false) RUN(8, 512, 16, false) RUN(8, 512, 24, false) RUN(16, 512, 16, false) RUN(32, 512, 16, false) RUN(8, 1024, 16, false)
  return 0;
}
