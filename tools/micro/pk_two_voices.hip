// Would two voices per lane in packed f32 (v_pk_mul/add/fma_f32) beat one voice per lane once the SIMDs are full?
//   The same recurrence -- the SVF step of svf.rs:262-279 as this library issues it (v3, two products
// of ic1, two of v3, the sums, the state update, the three-term output mix) followed by a gain and an envelope-like ramp --
// runs (A) one voice per lane in scalar f32 instructions and (B) two voices per lane, every operation a packed one over the
// two voices.  Per variant and occupancy (1, 2, 4 wavefronts per SIMD): voice-samples per second of the chip, and the
// instruction counts the compiler made of each (llvm-objdump of this file gives them).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o pk_two_voices pk_two_voices.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <typename V>
struct SvfState { V ic1, ic2, a1, a2, a3, m0, m1, m2, g, env, rate; };

template <typename V>
__device__ __forceinline__ V step(SvfState<V>& s, V x) {
  V v3 = x - s.ic2;
  V v1 = s.a1 * s.ic1 + s.a2 * v3;
  V v2 = (s.ic2 + s.a2 * s.ic1) + s.a3 * v3;
  s.ic1 = (v1 + v1) - s.ic1;
  s.ic2 = (v2 + v2) - s.ic2;
  V y = (s.m0 * x + s.m1 * v1) + s.m2 * v2;
  s.env = s.env + s.rate;
  return (y * s.g) * s.env;
}

template <typename V>
__device__ __forceinline__ V make(float a, float b);
template <> __device__ __forceinline__ float make<float>(float a, float) { return a; }
template <> __device__ __forceinline__ f2 make<f2>(float a, float b) { return f2{a, b}; }
__device__ __forceinline__ float total(float v) { return v; }
__device__ __forceinline__ float total(f2 v) { return v.x + v.y; }

template <typename V>
__global__ void __launch_bounds__(1024) k(float* out, int samples) {
  const float t = (threadIdx.x + blockIdx.x * blockDim.x) * 1e-6f;
  SvfState<V> s;
  s.ic1 = make<V>(0.f, 0.f); s.ic2 = make<V>(0.f, 0.f);
  s.a1 = make<V>(0.9f + t, 0.91f + t); s.a2 = make<V>(0.05f + t, 0.051f); s.a3 = make<V>(0.002f, 0.0021f + t);
  s.m0 = make<V>(0.f, 0.f); s.m1 = make<V>(0.f, 0.f); s.m2 = make<V>(1.f, 1.f);
  s.g = make<V>(0.5f, 0.4f); s.env = make<V>(0.f, 0.f); s.rate = make<V>(1e-6f, 2e-6f);
  V x = make<V>(0.3f + t, 0.2f - t), dx = make<V>(1e-3f, -1e-3f), acc = make<V>(0.f, 0.f);
  for (int i = 0; i < samples; i += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      x = x + dx;                 // (stands for the oscillator: one add; its table read has no packed form anyway)
      acc = acc + step(s, x);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = total(acc);
}

template <typename V>
double run(float* d, int waves_per_cu, int samples) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<V><<<256, 64 * waves_per_cu>>>(d, 64);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<V><<<256, 64 * waves_per_cu>>>(d, samples);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e-3;
}

int main() {
  float* d;
  (void)hipMalloc(&d, 256 * 1024 * 4);
  const int samples = 1 << 18;
  std::printf("%-34s %12s %14s %22s\n", "variant", "waves/SIMD", "ns per sample", "voice-samples/s (chip)");
  for (int w : {4, 8, 16}) {
    double ta = run<float>(d, w, samples), tb = run<f2>(d, w, samples);
    double va = 256.0 * 64 * w * samples / ta, vb = 256.0 * 64 * w * 2 * samples / tb;
    std::printf("%-34s %12d %14.2f %22.3e\n", "one voice per lane, scalar f32", w / 4, ta * 1e9 / samples, va);
    std::printf("%-34s %12d %14.2f %22.3e   x%.2f\n", "two voices per lane, packed f32", w / 4, tb * 1e9 / samples, vb, vb / va);
  }
  return 0;
}
