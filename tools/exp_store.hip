// Experiment: attainable HBM write bandwidth on MI355X for the tile-store patterns of the Gram build.
//   hipcc --offload-arch=gfx950 -O3 tools/exp_store.hip -o /tmp/exp_store && /tmp/exp_store
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double dv2 __attribute__((ext_vector_type(2)));

// pattern A: each workgroup writes TR rows x TC doubles (tile), tiles row-major over an N x N matrix
template <int TR, int TC, bool NT>
__global__ __launch_bounds__(256) void tile_store(double* K, long long N, long long tiles_per_row) {
  const long long t = blockIdx.x;
  const long long ti = t / tiles_per_row, tj = t % tiles_per_row;
  constexpr int LPR = TC / 2;            // lanes per row (16 B each)
  constexpr int RPP = 256 / LPR;         // rows per pass
  const int lp = threadIdx.x % LPR, r0 = threadIdx.x / LPR;
  for (int r = r0; r < TR; r += RPP) {
    double* p = K + (ti * TR + r) * N + tj * TC + 2 * lp;
    dv2 v = {(double)r, (double)lp};
    if (NT) __builtin_nontemporal_store(v, (dv2*)p); else *(dv2*)p = v;
  }
}
// pattern B: fully contiguous streaming store
__global__ __launch_bounds__(256) void stream_store(double* K, long long n2) {
  long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 2;
  const long long stride = (long long)gridDim.x * 256 * 2;
  for (; i < n2; i += stride) { dv2 v = {1.0, 2.0}; *(dv2*)(K + i) = v; }
}
template <typename F> float timeit(F f, int it = 5) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  float best = 1e30f;
  for (int i = 0; i < it; ++i) { hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
  return best;
}
int main() {
  const long long N = 65536; double* K; hipMalloc(&K, N * N * 8);
  const double gb = N * N * 8 / 1e9;
  float ms;
  ms = timeit([&] { hipLaunchKernelGGL(stream_store, dim3(256 * 16), dim3(256), 0, 0, K, N * N); });
  printf("stream (grid-stride, 4096 WGs):        %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
  ms = timeit([&] { hipLaunchKernelGGL(stream_store, dim3(256 * 64), dim3(256), 0, 0, K, N * N); });
  printf("stream (grid-stride, 16384 WGs):       %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
#define RUN(TR, TC, NT) ms = timeit([&] { hipLaunchKernelGGL((tile_store<TR, TC, NT>), dim3((unsigned)((N / TR) * (N / TC))), dim3(256), 0, 0, K, N, N / TC); }); \
  printf("tile %3d x %4d doubles (%5d B rows) nt=%d: %.3f ms  %.0f GB/s\n", TR, TC, TC * 8, (int)NT, ms, gb / ms * 1e3);
  RUN(64, 64, false) RUN(64, 64, true) RUN(64, 128, false) RUN(32, 128, false) RUN(64, 256, false) RUN(16, 256, false)
  RUN(128, 128, false) RUN(8, 512, false) RUN(4, 1024, false) RUN(1, 4096, false) RUN(64, 32, false) RUN(128, 16, false)
  return 0;
}
