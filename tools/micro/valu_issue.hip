// Issue cost of scalar vs packed FP32 VALU for a wavefront that is alone on its SIMD (gfx950).
// Each variant runs `iters` x 16 instructions; "dep" = every instruction reads the previous result,
// "ind" = 8 independent accumulator chains interleaved.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2_ __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(float* out, int iters) {
  float a0 = threadIdx.x * 1e-3f + 1.0f, b = 0.9999f;
  float a[8];
  float2_ p[8];
  float2_ pb = {0.9999f, 0.9998f};
  for (int j = 0; j < 8; ++j) { a[j] = a0 + j; p[j] = float2_{a0 + j, a0 - j}; }
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {  // scalar, dependent
#pragma unroll
      for (int u = 0; u < 16; ++u) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[0]) : "v"(b));
    } else if (MODE == 1) {  // scalar, independent
#pragma unroll
      for (int u = 0; u < 16; ++u) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[u & 7]) : "v"(b));
    } else if (MODE == 2) {  // packed, dependent
#pragma unroll
      for (int u = 0; u < 16; ++u) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[0]) : "v"(pb));
    } else if (MODE == 3) {  // packed, independent
#pragma unroll
      for (int u = 0; u < 16; ++u) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[u & 7]) : "v"(pb));
    } else if (MODE == 4) {  // scalar fma dependent
#pragma unroll
      for (int u = 0; u < 16; ++u) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[0]) : "v"(b));
    } else if (MODE == 5) {  // scalar, pairs: two independent chains alternating (distance 2)
#pragma unroll
      for (int u = 0; u < 16; ++u) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[u & 1]) : "v"(b));
    } else if (MODE == 6) {  // packed, pairs
#pragma unroll
      for (int u = 0; u < 16; ++u) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[u & 1]) : "v"(pb));
    } else if (MODE == 9) {  // scalar, independent, an s_nop 0 after every instruction
#pragma unroll
      for (int u = 0; u < 16; ++u) asm volatile("v_mul_f32 %0, %0, %1\n\ts_nop 0" : "+v"(a[u & 7]) : "v"(b));
    } else if (MODE == 10) {  // packed dependent with the s_nop 0 the hazard recogniser inserts
#pragma unroll
      for (int u = 0; u < 16; ++u) asm volatile("v_pk_mul_f32 %0, %0, %1\n\ts_nop 0" : "+v"(p[0]) : "v"(pb));
    } else if (MODE == 11) {  // 10-instruction SVF-like mix: 5 packed + 5 scalar, one dependent chain of 4
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        asm volatile(
            "v_sub_f32 %0, %4, %1\n\t"
            "v_pk_mul_f32 %2, %6, %3\n\t"
            "v_mul_f32 %5, %4, %4\n\t"
            "v_pk_mul_f32 %3, %6, %2\n\t"
            "v_add_f32 %4, %4, %5\n\t"
            "v_pk_add_f32 %2, %2, %3\n\t"
            "v_add_f32 %5, %5, %4\n\t"
            "v_pk_fma_f32 %3, %2, %6, %3\n\t"
            : "+v"(a[0]), "+v"(a[1]), "+v"(p[0]), "+v"(p[1]), "+v"(a[2]), "+v"(a[3]) : "v"(pb));
      }
    } else if (MODE == 12) {  // scalar fma, independent
#pragma unroll
      for (int u = 0; u < 16; ++u) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[u & 7]) : "v"(b));
    } else if (MODE == 13) {  // v_add_f32 (VOP2) independent
#pragma unroll
      for (int u = 0; u < 16; ++u) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[u & 7]) : "v"(b));
    } else if (MODE == 14) {  // packed fma, independent
#pragma unroll
      for (int u = 0; u < 16; ++u) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[u & 7]) : "v"(pb));
    } else if (MODE == 15) {  // packed add, independent
#pragma unroll
      for (int u = 0; u < 16; ++u) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[u & 7]) : "v"(pb));
    } else if (MODE == 7) {  // f64 mul dependent
      double d = a[0];
#pragma unroll
      for (int u = 0; u < 16; ++u) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d) : "v"((double)b));
      a[0] = (float)d;
    } else if (MODE == 8) {  // f64 mul independent
      double d[8];
      for (int j = 0; j < 8; ++j) d[j] = a[j];
#pragma unroll
      for (int u = 0; u < 16; ++u) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[u & 7]) : "v"((double)b));
      for (int j = 0; j < 8; ++j) a[j] = (float)d[j];
    }
  }
  float s = 0;
  for (int j = 0; j < 8; ++j) s += a[j] + p[j].x + p[j].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, float* d, int waves) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  k<MODE><<<256, 64 * waves>>>(d, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<256, 64 * waves>>>(d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  std::printf("%-34s waves/CU %d: %.3f ns per instruction\n", name, waves, ms * 1e6 / (iters * 16.0));
}

int main() {
  float* d;
  (void)hipMalloc(&d, 256 * 1024 * 4);
  for (int waves : {1, 4, 8, 16}) {
    run<0>("v_mul_f32 dependent", d, waves);
    run<5>("v_mul_f32 two chains", d, waves);
    run<1>("v_mul_f32 eight chains", d, waves);
    run<4>("v_fma_f32 dependent", d, waves);
    run<12>("v_fma_f32 eight chains", d, waves);
    run<13>("v_add_f32 eight chains", d, waves);
    run<2>("v_pk_mul_f32 dependent", d, waves);
    run<6>("v_pk_mul_f32 two chains", d, waves);
    run<3>("v_pk_mul_f32 eight chains", d, waves);
    run<15>("v_pk_add_f32 eight chains", d, waves);
    run<14>("v_pk_fma_f32 eight chains", d, waves);
    run<9>("v_mul_f32 eight chains + s_nop 0", d, waves);
    run<10>("v_pk_mul_f32 dependent + s_nop 0", d, waves);
    run<11>("8-instr pk/scalar mix (per instr)", d, waves);
    run<7>("v_mul_f64 dependent", d, waves);
    run<8>("v_mul_f64 eight chains", d, waves);
  }
  return 0;
}
