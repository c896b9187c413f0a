// Does a wave64 whose EXEC mask covers only 16 or 32 lanes issue VALU faster than a full wave on gfx950?
// (decides whether 16-voice waves could quadruple the SIMD count usable at 16 384 voices)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void chain(float* out, int active, int iters) {
  const int lane = threadIdx.x & 63;
  float a = 1.0f + lane * 1e-3f, b = 0.999f, c = 1e-4f;
  if (lane < active) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int k = 0; k < 16; ++k) a = __builtin_fmaf(a, b, c);
    }
    out[blockIdx.x * 64 + lane] = a;
  }
}
int main() {
  float* d;
  hipMalloc(&d, 256 * 64 * 4 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int waves = 1; waves <= 2; ++waves)
    for (int active : {64, 48, 32, 16, 8}) {
      chain<<<256, 64 * waves>>>(d, active, 100);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      chain<<<256, 64 * waves>>>(d, active, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      std::printf("waves/block %d active lanes %2d: %.3f ms  -> %.2f ns per dependent FMA\n", waves, active, ms, ms * 1e6 / (iters * 16.0));
    }
  return 0;
}
