// Experiment: which XCD does workgroup b of a 1-D grid run on?  (HW_REG_XCC_ID, MI355X_MICROARCH.md)
//   hipcc --offload-arch=gfx950 -O3 tools/exp_xcc.hip -o /tmp/exp_xcc && /tmp/exp_xcc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void who(int* out, int spin) {
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
  // keep the block alive a little so that residency resembles a real kernel
  float x = threadIdx.x;
  for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;
  if (threadIdx.x == 0) out[blockIdx.x] = (int)(id & 0xf) + (x < 0 ? 1 : 0);
}
int main() {
  for (int threads : {256, 512}) {
    for (int nb : {4096, 33792}) {
      int* d; hipMalloc(&d, nb * sizeof(int));
      hipLaunchKernelGGL(who, dim3(nb), dim3(threads), 0, 0, d, 20000);
      std::vector<int> h(nb); hipMemcpy(h.data(), d, nb * sizeof(int), hipMemcpyDeviceToHost);
      int match = 0, hist[16] = {0};
      for (int b = 0; b < nb; ++b) { match += (h[b] == h[b % 8]); hist[h[b] & 15]++; }
      printf("threads=%d blocks=%d: first 16 xcc ids:", threads, nb);
      for (int b = 0; b < 16; ++b) printf(" %d", h[b]);
      printf(" | blocks with xcc(b)==xcc(b%%8): %d/%d | per-xcc counts:", match, nb);
      for (int x = 0; x < 8; ++x) printf(" %d", hist[x]);
      printf("\n");
      hipFree(d);
    }
  }
  return 0;
}
