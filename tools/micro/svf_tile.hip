// The filter wave's tile arithmetic as the pipeline kernel compiles it (Svf::tick_tile_packed<64> from voice_chain.hpp: tile in
// from LDS rows, 64 steps, tile out to LDS rows), timed with s_memtime in a wavefront of its own while the other three
// wavefronts of the workgroup (a) idle at the barrier, (b) run an unrelated VALU loop, (c) gather from a 64 KiB LDS table
// like the oscillator wave, (d) run the filter themselves.  Answers: is the 52 cycles per sample seen inside the pipeline
// kernel (against 42 in tools/micro/svf_chain.hip) the code around the steps, or the neighbours?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I. -o tools/micro/svf_tile tools/micro/svf_tile.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include "knaster_amd/csrc/voice_chain.hpp"
using namespace knh_dev;

template <int OTHERS>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* ticks, int tiles) {
  __shared__ float table[16384];
  __shared__ __attribute__((aligned(16))) float rows[2][64][68];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16384; i += 256) table[i] = __sinf(i * 3.8349519e-4f);
  for (int i = threadIdx.x; i < 2 * 64 * 68; i += 256) (&rows[0][0][0])[i] = __sinf(0.01f * i);
  __syncthreads();
  Svf::Regs<float> r;
  r.ic1 = 0.0f; r.ic2 = 0.0f; r.a1 = 0.98f; r.a2 = 0.07f; r.a3 = 0.005f; r.m0 = 0.1f; r.m1 = 0.2f; r.m2 = 1.0f;
  unsigned long long busy = 0, t_in = 0, t_st = 0, t_drain = 0, t_bar = 0;
  float acc = lane * 1e-3f;
  unsigned ph = lane * 2654435761u;
  unsigned long long tb0 = 0;
  float yp[64], yn[64];
  for (int k = 0; k < 64; ++k) { yp[k] = 0.0f; yn[k] = 0.0f; }
  for (int t = 0; t < tiles; ++t) {
    if (wave != 0 && OTHERS != 3) tb0 = 0;
    if (wave == 0 || OTHERS == 3) {
      const unsigned long long t0 = __builtin_amdgcn_s_memtime();
      float x[64];
      typedef float V4 __attribute__((ext_vector_type(4)));
      const V4* in = reinterpret_cast<const V4*>(&rows[t & 1][lane][0]);
#pragma unroll
      for (int j = 0; j < 16; ++j) { const V4 v = in[j]; x[4 * j] = v[0]; x[4 * j + 1] = v[1]; x[4 * j + 2] = v[2]; x[4 * j + 3] = v[3]; }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const unsigned long long t1 = __builtin_amdgcn_s_memtime();
      V4* o = reinterpret_cast<V4*>(&rows[(t + 1) & 1][lane][0]);
      if (OTHERS == 6) {
        // the previous tile's results (yp) go out two stores at a time between the runs of eight steps of this tile
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          float xc[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) xc[k] = x[8 * c + k];
          asm volatile("" : "+v"(xc[0]), "+v"(xc[1]), "+v"(xc[2]), "+v"(xc[3]), "+v"(xc[4]), "+v"(xc[5]), "+v"(xc[6]), "+v"(xc[7]));
          Svf::tick_tile_packed<8>(r, xc);
          V4 v0 = {yp[8 * c], yp[8 * c + 1], yp[8 * c + 2], yp[8 * c + 3]}, v1 = {yp[8 * c + 4], yp[8 * c + 5], yp[8 * c + 6], yp[8 * c + 7]};
          o[2 * c] = v0;
          o[2 * c + 1] = v1;
#pragma unroll
          for (int k = 0; k < 8; ++k) yn[8 * c + k] = xc[k];
        }
#pragma unroll
        for (int k = 0; k < 64; ++k) yp[k] = yn[k];
      } else {
        Svf::tick_tile_packed<64>(r, x);
      }
      asm volatile("" ::: "memory");
      const unsigned long long t2 = __builtin_amdgcn_s_memtime();
      if (OTHERS != 6) {
#pragma unroll
        for (int j = 0; j < 16; ++j) { V4 v = {x[4 * j], x[4 * j + 1], x[4 * j + 2], x[4 * j + 3]}; o[j] = v; }
      }
      const unsigned long long t3 = __builtin_amdgcn_s_memtime();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const unsigned long long t4 = __builtin_amdgcn_s_memtime();
      if (wave == 0) { busy += t2 - t1; t_in += t1 - t0; t_st += t3 - t2; t_drain += t4 - t3; tb0 = t4; }
    } else if (OTHERS == 1) {
      for (int i = 0; i < 600; ++i) acc = acc * 0.9999f + 1e-4f;
    } else if (OTHERS == 2) {
      for (int i = 0; i < 64; ++i) { acc += table[(ph >> 18) & 16383u]; ph += 0x01234567u; }
    } else if (OTHERS == 4) {
      // straight-line code, a different 6 KiB of it in each wavefront (like the unrolled tiles of the oscillator and envelope
      // groups): ~2 600 cycles of independent VALU work per tile
      if (wave == 1) {
#pragma unroll
        for (int i = 0; i < 768; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc) : "v"(0.9999f), "v"(1e-4f + i));
      } else if (wave == 2) {
#pragma unroll
        for (int i = 0; i < 768; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc) : "v"(0.9998f), "v"(2e-4f + i));
      } else {
#pragma unroll
        for (int i = 0; i < 768; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc) : "v"(0.9997f), "v"(3e-4f + i));
      }
    } else if (OTHERS == 5) {
      // the same amount of work as a loop (one 64-byte line of code)
      for (int i = 0; i < 768; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc) : "v"(0.9999f), "v"(1e-4f));
    }
    __syncthreads();
    if (wave == 0) t_bar += __builtin_amdgcn_s_memtime() - tb0;
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc + r.ic1 + r.ic2;
  if (threadIdx.x == 0 && blockIdx.x == 3) { ticks[0] = busy; ticks[1] = t_in; ticks[2] = t_st; ticks[3] = t_drain; ticks[4] = t_bar; }
}

template <int OTHERS>
void run(const char* name, float* d) {
  unsigned long long* ticks;
  (void)hipHostMalloc(&ticks, 64);
  const int tiles = 20000;
  k<OTHERS><<<256, 256>>>(d, ticks, 200);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  k<OTHERS><<<256, 256>>>(d, ticks, tiles);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  std::printf("%-44s filter steps: %7.2f s_memtime ticks per sample; whole tile step %.1f ns (%.2f ns per sample)\n", name,
              (double)ticks[0] / (tiles * 64.0), ms * 1e6 / tiles, ms * 1e6 / tiles / 64.0);
  std::printf("    per tile: tile in (16 reads + wait) %.0f, steps %.0f, 16 stores issued %.0f, their drain %.0f, barrier (+ loop) %.0f ticks\n",
              (double)ticks[1] / tiles, (double)ticks[0] / tiles, (double)ticks[2] / tiles, (double)ticks[3] / tiles, (double)ticks[4] / tiles);
  (void)hipHostFree(ticks);
}

int main() {
  float* d;
  (void)hipMalloc(&d, 256 * 256 * 4);
  run<0>("others idle at the barrier", d);
  run<1>("others run a VALU loop", d);
  run<2>("others gather from the LDS table", d);
  run<3>("all four wavefronts run the filter", d);
  run<4>("others run 6 KiB of straight-line VALU each", d);
  run<5>("others run the same work as a loop", d);
  run<6>("idle others; stores of tile k-1 between the runs of tile k", d);
  run<0>("others idle at the barrier (again)", d);
  return 0;
}
