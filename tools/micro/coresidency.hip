// Can a one-workgroup-per-CU kernel (256 threads, ~150 KiB of LDS, most of the register file: the resident voice kernel's shape)
// start on EVERY CU while a few single-wavefront workgroups of another kernel (the fold server's shape: no LDS, few registers)
// are already spinning on some of them?  Each big workgroup records when it started; the spinners run for 20 ms.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

__global__ void __launch_bounds__(64) spinner(uint64_t* started, uint64_t ticks) {
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) started[blockIdx.x] = t0;
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
template <int REGS>
__global__ void __launch_bounds__(256) big(uint64_t* started, float* sink, int n) {
  extern __shared__ float lds[];
  if (threadIdx.x == 0) started[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
  float r[REGS];  // keeps REGS registers live
#pragma unroll
  for (int k = 0; k < REGS; ++k) r[k] = (float)(threadIdx.x * (k + 1));
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int k = 0; k < REGS; ++k) r[k] = r[k] * 1.0001f + r[(k + 1) % REGS];
  }
  float s = 0;
#pragma unroll
  for (int k = 0; k < REGS; ++k) s += r[k];
  lds[threadIdx.x] = s;
  __syncthreads();
  if (s == 12345.f) sink[0] = lds[(threadIdx.x + 1) & 255];
}
int main() {
  int n_cu = 0;
  CK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0));
  uint64_t *d_sp = nullptr, *d_big = nullptr;
  float* sink = nullptr;
  CK(hipMalloc(&d_sp, 64 * 8));
  CK(hipMalloc(&d_big, 1024 * 8));
  CK(hipMalloc(&sink, 64));
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(big<280>), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(big<100>), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
  for (int regs = 0; regs < 2; ++regs) {
    for (int n_spin : {0, 9, 64}) {
      CK(hipMemset(d_big, 0, 1024 * 8));
      CK(hipMemset(d_sp, 0, 64 * 8));
      CK(hipDeviceSynchronize());
      if (n_spin) hipLaunchKernelGGL(spinner, dim3(n_spin), dim3(64), 0, s1, d_sp, (uint64_t)2000000);  // 20 ms
      if (regs) hipLaunchKernelGGL(big<280>, dim3(n_cu), dim3(256), 159 * 1024, s2, d_big, sink, 4);
      else hipLaunchKernelGGL(big<100>, dim3(n_cu), dim3(256), 159 * 1024, s2, d_big, sink, 4);
      CK(hipGetLastError());
      CK(hipDeviceSynchronize());
      std::vector<uint64_t> b(n_cu), sp(64);
      CK(hipMemcpy(b.data(), d_big, n_cu * 8, hipMemcpyDeviceToHost));
      CK(hipMemcpy(sp.data(), d_sp, 64 * 8, hipMemcpyDeviceToHost));
      const uint64_t first = *std::min_element(b.begin(), b.end()), last = *std::max_element(b.begin(), b.end());
      int late = 0;
      for (uint64_t t : b) late += (t - first) > 100000;  // started more than 1 ms after the first
      std::printf("big kernel with ~%d live registers per lane, %d spinning wavefronts first: %d workgroups, first-to-last start %.1f us, %d started late (after the spinners ended?)\n",
                  regs ? 280 : 100, n_spin, n_cu, (last - first) / 100.0, late);
    }
  }
  return 0;
}
