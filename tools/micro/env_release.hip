// Cycles per 64-sample tile of the envelope's release arithmetic, x[j] *= (t*(t*t))*scale; t += step, for a wavefront
// alone on its SIMD, in several formulations (what the compiler makes of each): the plain loop, the t chain computed
// first, the chain through opaque single adds, two samples per packed multiply written out by hand.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o env_release env_release.hip && ./env_release
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int T = 64;

template <int MODE>
__device__ __forceinline__ void tile(float (&x)[T], float& t, float step, float scale) {
  if (MODE == 0) {  // the shipped loop
#pragma unroll
    for (int j = 0; j < T; ++j) { const float tj = t; t = t + step; x[j] = x[j] * ((tj * (tj * tj)) * scale); }
  } else if (MODE == 1) {  // attack-style: x *= t
#pragma unroll
    for (int j = 0; j < T; ++j) { const float tj = t; t = t + step; x[j] = x[j] * tj; }
  } else if (MODE == 2) {  // the adds as opaque instructions (the compiler cannot re-associate or vectorise them)
#pragma unroll
    for (int j = 0; j < T; ++j) {
      const float tj = t;
      asm("v_add_f32 %0, %1, %2" : "=v"(t) : "v"(tj), "v"(step));
      x[j] = x[j] * ((tj * (tj * tj)) * scale);
    }
  } else if (MODE == 3) {  // explicit pairs
    const f2 sc = {scale, scale};
#pragma unroll
    for (int j = 0; j < T; j += 2) {
      const float ta = t, tb = ta + step;
      t = tb + step;
      const f2 tp = {ta, tb};
      f2 xv = {x[j], x[j + 1]};
      xv = xv * ((tp * (tp * tp)) * sc);
      x[j] = xv.x; x[j + 1] = xv.y;
    }
  } else if (MODE == 4) {  // no envelope arithmetic at all: x *= scale (the sustain tile)
#pragma unroll
    for (int j = 0; j < T; ++j) x[j] = x[j] * scale;
  }
}

template <int MODE>
__global__ void k(float* out, unsigned long long* ticks, int iters) {
  __shared__ __attribute__((aligned(16))) float tile_in[64 * 68], tile_out[64 * 68];
  const int lane = threadIdx.x;
  for (int i = lane; i < 64 * 68; i += 64) tile_in[i] = 0.001f * i;
  __syncthreads();
  float t = 1.0f, step = -1e-6f, scale = 0.7f;
  unsigned long long sum = 0;
  for (int it = 0; it < iters; ++it) {
    float x[T];
    const f4* in = reinterpret_cast<const f4*>(tile_in + lane * 68);
#pragma unroll
    for (int j = 0; j < T / 4; ++j) { const f4 v = in[j]; x[4 * j] = v.x; x[4 * j + 1] = v.y; x[4 * j + 2] = v.z; x[4 * j + 3] = v.w; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    tile<MODE>(x, t, step, scale);
    asm volatile("" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    sum += t1 - t0;
    f4* o = reinterpret_cast<f4*>(tile_out + lane * 68);
#pragma unroll
    for (int j = 0; j < T / 4; ++j) o[j] = f4{x[4 * j], x[4 * j + 1], x[4 * j + 2], x[4 * j + 3]};
    if (t < 0.5f) t = 1.0f;
  }
  out[blockIdx.x * 64 + lane] = tile_out[lane * 68 + 5] + t;
  if (lane == 0 && blockIdx.x == 3) ticks[0] = sum;
}

// whole-kernel time per tile (the stamps inside only bracket what the compiler leaves between them): each tile also
// reads and writes its 64 x 64 samples through LDS, the same in every variant -- compare against the sustain line
template <int MODE>
void run(const char* name, float* d, unsigned long long* ticks) {
  const int iters = 200000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<MODE><<<256, 64>>>(d, ticks, 1000);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<MODE><<<256, 64>>>(d, ticks, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  std::printf("%-58s %7.1f ns per 64-sample tile (x 2.4 = %5.0f cycles)\n", name, ms * 1e6 / iters, ms * 1e6 / iters * 2.4);
}

int main() {
  float* d;
  unsigned long long* ticks;
  (void)hipMalloc(&d, 256 * 64 * 4);
  (void)hipHostMalloc(&ticks, 8);
  run<4>("x *= scale (sustain)", d, ticks);
  run<1>("x *= t; t += step (attack)", d, ticks);
  run<0>("x *= (t*(t*t))*scale; t += step (release, as shipped)", d, ticks);
  run<2>("the same, t chain through opaque v_add_f32", d, ticks);
  run<3>("the same, sample pairs written as packed multiplies", d, ticks);
  return 0;
}
