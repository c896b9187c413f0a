// Does an f64 VALU instruction of a wave64 whose EXEC mask covers only 16 or 32 lanes hold the SIMD for less time than a
// full one (an f64 op takes four cycles for 64 lanes: would empty quarter-passes be skipped)?  Independent instructions
// (issue-bound), one wavefront per SIMD.  Decides whether f64 banks of few voices gain from 32-voice wavefronts.
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename F>
__global__ void indep(F* out, int active, int iters) {
  const int lane = threadIdx.x & 63;
  F a[8];
  for (int k = 0; k < 8; ++k) a[k] = (F)1 + (F)(lane + k) * (F)1e-3;
  const F b = (F)0.999, c = (F)1e-4;
  if (lane < active) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = a[k] * b + c;  // eight independent chains: each instruction reads a result 8 back
    }
    F s = 0;
    for (int k = 0; k < 8; ++k) s += a[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  }
}
template <typename F> void run(const char* name) {
  F* d;
  hipMalloc(&d, 256 * 512 * sizeof(F));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int waves : {4, 8})
    for (int active : {64, 48, 32, 16}) {
      indep<F><<<256, 64 * waves>>>(d, active, 100);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      indep<F><<<256, 64 * waves>>>(d, active, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      std::printf("%s waves/CU %d active lanes %2d: %.3f ms -> %.2f ns per fma (x2 ops: mul+add counted as one fma-shaped step)\n", name, waves, active, ms, ms * 1e6 / (iters * 16.0));
    }
  hipFree(d);
}
int main() { run<float>("f32"); run<double>("f64"); return 0; }
