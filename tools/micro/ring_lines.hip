// What the memory system makes of the delay rings' access pattern, without any of the voice arithmetic.
// Every voice owns a ring of `len` f32 samples ([voice][len], 48 000 bytes apart); per 32-sample tile it reads 128 bytes at
// its read position and writes 128 bytes at its write position, tile after tile.  Two ways to issue that from a wavefront
// of 64 voices:
//   lane-per-voice : every lane moves its own voice's 128 bytes as 8 x 16-byte accesses (64 lines per instruction, 16 bytes
//                    of each) -- what SampleDelay::tick_tile does
//   line-per-8-lanes: eight lanes move one voice's 128 contiguous bytes in one instruction (8 voices = 8 whole lines per
//                    instruction), eight instructions per tile
// Both with the next tile's loads issued ahead of this tile's stores.  Reports bytes moved per second (read + written).
//   hipcc --offload-arch=gfx950 -O3 -o ring_lines ring_lines.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4), aligned(4)));

template <bool COOP>
__global__ void __launch_bounds__(256) k(float* rings, unsigned len, unsigned n_voices, unsigned tiles, unsigned delay_skew, unsigned write_skew) {
  const unsigned lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const unsigned v0 = wave * 64u;
  if (v0 >= n_voices) return;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  if (!COOP) {
    const unsigned v = v0 + lane;
    float* ring = rings + (size_t)v * len;
    unsigned wp = (v * write_skew) % 32u, rp = (len - 4096u - (v * delay_skew) % 4096u) % len;   // a per-voice delay of 4 096 .. 8 191 samples
    for (unsigned t = 0; t < tiles; ++t) {
      f4 y[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) y[j] = *reinterpret_cast<const f4*>(ring + rp + 4 * j);
#pragma unroll
      for (int j = 0; j < 8; ++j) { acc += y[j]; *reinterpret_cast<f4*>(ring + wp + 4 * j) = y[j] + acc; }
      wp += 32u; if (wp + 32u > len) wp &= 31u;
      rp += 32u; if (rp + 32u > len) rp &= 31u;
    }
  } else {
    // instruction i moves chunk (lane & 7) of voices v0 + 8 i + (lane >> 3)
    unsigned wp[8], rp[8];
    float* ring[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const unsigned v = v0 + 8u * i + (lane >> 3);
      ring[i] = rings + (size_t)v * len + 4u * (lane & 7u);
      wp[i] = (v * write_skew) % 32u; rp[i] = (len - 4096u - (v * delay_skew) % 4096u) % len;
    }
    for (unsigned t = 0; t < tiles; ++t) {
      f4 y[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) y[i] = *reinterpret_cast<const f4*>(ring[i] + rp[i]);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        acc += y[i];
        *reinterpret_cast<f4*>(ring[i] + wp[i]) = y[i] + acc;
        wp[i] += 32u; if (wp[i] + 32u > len) wp[i] &= 31u;
        rp[i] += 32u; if (rp[i] + 32u > len) rp[i] &= 31u;
      }
    }
  }
  if (acc.x == 12345.f) rings[0] = acc.y;
}

int main(int argc, char** argv) {
  const unsigned len = 12000;
  std::printf("%-18s %9s %6s %12s %10s\n", "form", "voices", "skew", "us per tile", "GB/s");
  for (unsigned nv : {16384u, 65536u, 262144u}) {
    float* d;
    if (hipMalloc(&d, (size_t)nv * len * 4) != hipSuccess) { std::printf("no memory for %u voices\n", nv); continue; }
    (void)hipMemset(d, 0, (size_t)nv * len * 4);
    const unsigned tiles = 16 * 32;  // 32 blocks of 512
    for (unsigned mode : {0u, 1u, 7u, 100u}) {  // read skew 0: every read position on a line boundary; 1, 7: anywhere (4-byte aligned); 100: reads aligned, WRITES anywhere
      const unsigned skew = mode == 100u ? 0u : mode, wskew = mode == 100u ? 7u : 0u;
      for (int coop = 0; coop < 2; ++coop) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        const unsigned blocks = (nv / 64 + 3) / 4;
        for (int rep = 0; rep < 2; ++rep) {
          (void)hipEventRecord(e0);
          if (coop) k<true><<<blocks, 256>>>(d, len, nv, tiles, skew, wskew);
          else k<false><<<blocks, 256>>>(d, len, nv, tiles, skew, wskew);
          (void)hipEventRecord(e1);
          (void)hipEventSynchronize(e1);
        }
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double bytes = 2.0 * 128.0 * nv * tiles;
        std::printf("%-18s %9u %6u %12.3f %10.1f\n", coop ? "line-per-8-lanes" : "lane-per-voice", nv, mode, ms * 1e3 / tiles, bytes / (ms * 1e-3) / 1e9);
      }
    }
    (void)hipFree(d);
  }
  return 0;
}
