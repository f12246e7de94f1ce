// Experiment for a CU-partitioned look-ahead in the blocked Cholesky: do CU-masked streams
// (hipExtStreamCreateWithCUMask) partition the chip on this stack, and does a short kernel on a small partition run
// BESIDE a long kernel that fills the large one?
//   hipcc --offload-arch=gfx950 -O3 tools/exp_cumask.hip -o /tmp/exp_cumask && /tmp/exp_cumask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void busy(unsigned* where, long long spin, long long* clk) {
  unsigned xcc, hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  const long long t0 = wall_clock64();
  float x = threadIdx.x;
  for (long long i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;
  if (threadIdx.x == 0) {
    where[blockIdx.x] = ((xcc & 0xf) << 16) | ((hwid >> 8) & 0xf) | (((hwid >> 13) & 0x7) << 4) | (x < 0 ? 1u << 31 : 0u);   // xcc | se | cu
    clk[2 * blockIdx.x] = t0;
    clk[2 * blockIdx.x + 1] = wall_clock64();
  }
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  printf("CUs: %d\n", ncu);
  const int nwords = (ncu + 31) / 32;
  // partition: the last `small` CU bits for the side stream, the rest for the main stream
  for (int small : {8, 16, 32}) {
    std::vector<uint32_t> big_mask(nwords, 0), small_mask(nwords, 0);
    for (int c = 0; c < ncu; ++c) (c >= ncu - small ? small_mask : big_mask)[c / 32] |= 1u << (c % 32);
    hipStream_t sb, ss;
    hipError_t e1 = hipExtStreamCreateWithCUMask(&sb, nwords, big_mask.data());
    hipError_t e2 = hipExtStreamCreateWithCUMask(&ss, nwords, small_mask.data());
    if (e1 != hipSuccess || e2 != hipSuccess) { printf("hipExtStreamCreateWithCUMask: %s / %s\n", hipGetErrorString(e1), hipGetErrorString(e2)); return 1; }
    const int nbig = 8 * ncu, nsmall = 64;
    unsigned *wb, *ws; long long *cb, *cs;
    CK(hipMalloc(&wb, nbig * 4)); CK(hipMalloc(&ws, nsmall * 4)); CK(hipMalloc(&cb, nbig * 16)); CK(hipMalloc(&cs, nsmall * 16));
    hipLaunchKernelGGL(busy, dim3(64), dim3(256), 0, sb, wb, 1000, cb);   // warm-up
    hipLaunchKernelGGL(busy, dim3(64), dim3(256), 0, ss, ws, 1000, cs);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(busy, dim3(nbig), dim3(256), 0, sb, wb, 4000000, cb);    // long: several ms, 8 rounds of the partition
    hipLaunchKernelGGL(busy, dim3(nsmall), dim3(256), 0, ss, ws, 20000, cs);    // short, launched right behind it
    CK(hipDeviceSynchronize());
    std::vector<unsigned> hb(nbig), hs(nsmall); std::vector<long long> tb(2 * nbig), ts(2 * nsmall);
    CK(hipMemcpy(hb.data(), wb, nbig * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hs.data(), ws, nsmall * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(tb.data(), cb, nbig * 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(ts.data(), cs, nsmall * 16, hipMemcpyDeviceToHost));
    long long b0 = tb[0], b1 = tb[1], s0 = ts[0], s1 = ts[1];
    for (int i = 0; i < nbig; ++i) { if (tb[2 * i] < b0) b0 = tb[2 * i]; if (tb[2 * i + 1] > b1) b1 = tb[2 * i + 1]; }
    for (int i = 0; i < nsmall; ++i) { if (ts[2 * i] < s0) s0 = ts[2 * i]; if (ts[2 * i + 1] > s1) s1 = ts[2 * i + 1]; }
    // distinct (xcc, se, cu) triples each kernel ran on, and their overlap
    std::vector<unsigned> ub, us;
    auto uniq = [](std::vector<unsigned> v) { std::vector<unsigned> u; for (unsigned x : v) { x &= 0x7fffffffu; bool f = false; for (unsigned y : u) f |= (y == x); if (!f) u.push_back(x); } return u; };
    ub = uniq(hb); us = uniq(hs);
    int common = 0; for (unsigned x : us) for (unsigned y : ub) common += (x == y);
    printf("side partition %2d CUs: long kernel on %zu distinct (xcc,se,cu), short on %zu, in common %d | long ran %.2f ms, short started %.2f ms after the long "
           "one started and ended %.2f ms before it ended (100 MHz clock)\n", small, ub.size(), us.size(), common, (b1 - b0) * 1e-5, (s0 - b0) * 1e-5, (b1 - s1) * 1e-5);
    printf("   short kernel's xcc ids:"); for (int i = 0; i < 16; ++i) printf(" %u", hs[i] >> 16 & 0xf); printf("\n");
    hipFree(wb); hipFree(ws); hipFree(cb); hipFree(cs); hipStreamDestroy(sb); hipStreamDestroy(ss);
  }
  return 0;
}
