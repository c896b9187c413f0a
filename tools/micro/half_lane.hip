// Would two 32-voice wavefronts on ONE SIMD get through the filter role's tile (64 rows in from LDS, 64 steps, 64 rows out,
// barrier) faster than one 64-voice wavefront does?  A wavefront alone on its SIMD leaves it half idle (valu_issue.hip), its
// LDS stores stall its own issue (svf_tile.hip), and the low-pass step is bound by the latency of its dependent chain, not by
// issue -- all three are cycles a second instruction stream on the same SIMD could use.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I. -o tools/micro/half_lane tools/micro/half_lane.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include "knaster_amd/csrc/voice_chain.hpp"
using namespace knh_dev;

// LANES active lanes per filter wavefront; the filter wavefronts are wave 0 and (when SECOND >= 0) wave SECOND of 8
template <int LANES, int SECOND, bool LOW>
__global__ void __launch_bounds__(512) k(float* out, int tiles) {
  __shared__ __attribute__((aligned(16))) float rows[2][2][64][68];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 2 * 2 * 64 * 68; i += 512) (&rows[0][0][0][0])[i] = __sinf(0.01f * i);
  __syncthreads();
  Svf::Regs<float> r;
  r.ic1 = 0.0f; r.ic2 = 0.0f; r.a1 = 0.98f; r.a2 = 0.07f; r.a3 = 0.005f;
  r.m0 = LOW ? 0.0f : 0.1f; r.m1 = LOW ? 0.0f : 0.2f; r.m2 = 1.0f;
  const bool mine = (wave == 0 || wave == SECOND) && lane < LANES;
  const int w = wave == 0 ? 0 : 1;
  for (int t = 0; t < tiles; ++t) {
    if (mine) {
      float x[64];
      typedef float V4 __attribute__((ext_vector_type(4)));
      const V4* in = reinterpret_cast<const V4*>(&rows[w][t & 1][lane][0]);
#pragma unroll
      for (int j = 0; j < 16; ++j) { const V4 v = in[j]; x[4 * j] = v[0]; x[4 * j + 1] = v[1]; x[4 * j + 2] = v[2]; x[4 * j + 3] = v[3]; }
      if (LOW) Svf::tick_tile_low<64>(r, x); else Svf::tick_tile_packed<64>(r, x);
      V4* o = reinterpret_cast<V4*>(&rows[w][(t + 1) & 1][lane][0]);
#pragma unroll
      for (int j = 0; j < 16; ++j) { V4 v = {x[4 * j], x[4 * j + 1], x[4 * j + 2], x[4 * j + 3]}; o[j] = v; }
    }
    __syncthreads();
  }
  out[blockIdx.x * 512 + threadIdx.x] = r.ic1 + r.ic2;
}

template <int LANES, int SECOND, bool LOW>
void run(const char* name, float* d) {
  const int tiles = 20000;
  k<LANES, SECOND, LOW><<<256, 512>>>(d, 2000);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  k<LANES, SECOND, LOW><<<256, 512>>>(d, tiles);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const int voices = LANES * (SECOND >= 0 ? 2 : 1);
  std::printf("%-64s tile step %7.1f ns = %6.2f ns per sample for %3d voices\n", name, ms * 1e6 / tiles, ms * 1e6 / tiles / 64.0, voices);
}

int main() {
  float* d;
  (void)hipMalloc(&d, 256 * 512 * 4);
  run<64, -1, true>("low-pass step: one wavefront, 64 lanes", d);
  run<32, -1, true>("low-pass step: one wavefront, 32 lanes", d);
  run<32, 4, true>("low-pass step: waves 0 and 4 (one SIMD), 32 lanes each", d);
  run<32, 1, true>("low-pass step: waves 0 and 1 (two SIMDs), 32 lanes each", d);
  run<64, 4, true>("low-pass step: waves 0 and 4 (one SIMD), 64 lanes each", d);
  run<64, 1, true>("low-pass step: waves 0 and 1 (two SIMDs), 64 lanes each", d);
  run<64, -1, false>("general step: one wavefront, 64 lanes", d);
  run<32, 4, false>("general step: waves 0 and 4 (one SIMD), 32 lanes each", d);
  run<64, 4, false>("general step: waves 0 and 4 (one SIMD), 64 lanes each", d);
  run<64, -1, true>("low-pass step: one wavefront, 64 lanes (again)", d);
  return 0;
}
