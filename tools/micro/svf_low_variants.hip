// Encodings of the low-pass filter step (Svf::tick_tile_low) for a wavefront alone on its SIMD, s_memtime ticks per sample:
//   A  the shipped one: sub, pk_mul, pk_mul, add, pk_add, fma(out), pk_fma, s_nop
//   B  the state update as two scalar fmas instead of a packed one (no wait state left to fill: eight instructions, no s_nop)
//   D  A without its s_nop (7 slots): the packed fma's result read by the very next instruction
//   E  six instructions and a test per run of eight samples: the output IS v2 unless v1 stopped being finite in the run
//   C  A without the output instruction (two s_nops): what the recurrence alone costs (not a usable step: NaN inputs)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I. -o tools/micro/svf_low_variants tools/micro/svf_low_variants.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));

#define STEP_A(OUT, IN, V, PL, PH)                                                                                 \
      "v_sub_f32 v114, %[x" #IN "], v101\n\t"                                                                      \
      "v_pk_mul_f32 v[102:103], %[a12], v[100:101] op_sel_hi:[1,0]\n\t"                                            \
      "v_pk_mul_f32 v[104:105], %[a23], v[114:115] op_sel_hi:[1,0]\n\t"                                            \
      "v_add_f32 v103, v101, v103\n\t"                                                                             \
      "v_pk_add_f32 " V ", v[102:103], v[104:105]\n\t"                                                             \
      "v_fma_f32 %[y" #OUT "], 0, " PL ", " PH "\n\t"                                                              \
      "v_pk_fma_f32 v[100:101], " V ", 2.0, v[100:101] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"        \
      "s_nop 0\n\t"
#define STEP_D(OUT, IN, V, PL, PH)                                                                                 \
      "v_sub_f32 v114, %[x" #IN "], v101\n\t"                                                                      \
      "v_pk_mul_f32 v[102:103], %[a12], v[100:101] op_sel_hi:[1,0]\n\t"                                            \
      "v_pk_mul_f32 v[104:105], %[a23], v[114:115] op_sel_hi:[1,0]\n\t"                                            \
      "v_add_f32 v103, v101, v103\n\t"                                                                             \
      "v_pk_add_f32 " V ", v[102:103], v[104:105]\n\t"                                                             \
      "v_fma_f32 %[y" #OUT "], 0, " PL ", " PH "\n\t"                                                              \
      "v_pk_fma_f32 v[100:101], " V ", 2.0, v[100:101] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"
#define STEP_B(OUT, IN, V, VL, VH, PL, PH)                                                                         \
      "v_sub_f32 v114, %[x" #IN "], v101\n\t"                                                                      \
      "v_pk_mul_f32 v[102:103], %[a12], v[100:101] op_sel_hi:[1,0]\n\t"                                            \
      "v_pk_mul_f32 v[104:105], %[a23], v[114:115] op_sel_hi:[1,0]\n\t"                                            \
      "v_add_f32 v103, v101, v103\n\t"                                                                             \
      "v_pk_add_f32 " V ", v[102:103], v[104:105]\n\t"                                                             \
      "v_fma_f32 %[y" #OUT "], 0, " PL ", " PH "\n\t"                                                              \
      "v_fma_f32 v101, 2.0, " VH ", -v101\n\t"                                                                     \
      "v_fma_f32 v100, 2.0, " VL ", -v100\n\t"
#define STEP_C(OUT, IN, V, PL, PH)                                                                                 \
      "v_sub_f32 v114, %[x" #IN "], v101\n\t"                                                                      \
      "v_pk_mul_f32 v[102:103], %[a12], v[100:101] op_sel_hi:[1,0]\n\t"                                            \
      "v_pk_mul_f32 v[104:105], %[a23], v[114:115] op_sel_hi:[1,0]\n\t"                                            \
      "v_add_f32 v103, v101, v103\n\t"                                                                             \
      "v_pk_add_f32 " V ", v[102:103], v[104:105]\n\t"                                                             \
      "s_nop 0\n\t"                                                                                                \
      "v_mov_b32 %[y" #OUT "], " PH "\n\t"                                                                         \
      "v_pk_fma_f32 v[100:101], " V ", 2.0, v[100:101] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"        \
      "s_nop 0\n\t"

#define STEP_E(K)                                                                                                  \
      "v_sub_f32 v114, %[x" #K "], v101\n\t"                                                                       \
      "v_pk_mul_f32 v[102:103], %[a12], v[100:101] op_sel_hi:[1,0]\n\t"                                            \
      "v_pk_mul_f32 v[104:105], %[a23], v[114:115] op_sel_hi:[1,0]\n\t"                                            \
      "v_add_f32 v103, v101, v103\n\t"                                                                             \
      "v_pk_add_f32 %[p" #K "], v[102:103], v[104:105]\n\t"                                                        \
      "v_pk_fma_f32 v[100:101], %[p" #K "], 2.0, v[100:101] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"
__device__ __forceinline__ void tile_e(float& ic1, float& ic2, f2 a12, f2 a23, float (&x)[64]) {
  f2 ic = {ic1, ic2};
#pragma unroll
  for (int j = 0; j < 64; j += 8) {
    f2 p0, p1, p2, p3, p4, p5, p6, p7;
    asm volatile(STEP_E(0) STEP_E(1) STEP_E(2) STEP_E(3) STEP_E(4) STEP_E(5) STEP_E(6) STEP_E(7)
                 : [p0] "=&v"(p0), [p1] "=&v"(p1), [p2] "=&v"(p2), [p3] "=&v"(p3), [p4] "=&v"(p4), [p5] "=&v"(p5), [p6] "=&v"(p6), [p7] "=&v"(p7),
                   "+{v[100:101]}"(ic)
                 : [x0] "v"(x[j]), [x1] "v"(x[j + 1]), [x2] "v"(x[j + 2]), [x3] "v"(x[j + 3]), [x4] "v"(x[j + 4]), [x5] "v"(x[j + 5]), [x6] "v"(x[j + 6]),
                   [x7] "v"(x[j + 7]), [a12] "v"(a12), [a23] "v"(a23)
                 : "v102", "v103", "v104", "v105", "v114", "v115");
    x[j] = p0.y; x[j + 1] = p1.y; x[j + 2] = p2.y; x[j + 3] = p3.y; x[j + 4] = p4.y; x[j + 5] = p5.y; x[j + 6] = p6.y; x[j + 7] = p7.y;
    // v1 not finite in some sample of the run -> ic1 not finite after it (it stays so): only then does the reference's 0 * v1 matter
    if (__builtin_amdgcn_ballot_w64(!(__builtin_fabsf(ic.x) < __builtin_inff())) != 0) {
      x[j] = __builtin_fmaf(0.0f, p0.x, p0.y); x[j + 1] = __builtin_fmaf(0.0f, p1.x, p1.y); x[j + 2] = __builtin_fmaf(0.0f, p2.x, p2.y);
      x[j + 3] = __builtin_fmaf(0.0f, p3.x, p3.y); x[j + 4] = __builtin_fmaf(0.0f, p4.x, p4.y); x[j + 5] = __builtin_fmaf(0.0f, p5.x, p5.y);
      x[j + 6] = __builtin_fmaf(0.0f, p6.x, p6.y); x[j + 7] = __builtin_fmaf(0.0f, p7.x, p7.y);
    }
  }
  ic1 = ic.x; ic2 = ic.y;
}

template <int VAR>
__device__ __forceinline__ void tile(float& ic1, float& ic2, f2 a12, f2 a23, float (&x)[64]) {
  if (VAR == 4) { tile_e(ic1, ic2, a12, a23, x); return; }
  f2 ic = {ic1, ic2};
  f2 q = {0.0f, 0.0f};
#pragma unroll
  for (int j = 0; j < 64; j += 8) {
    float y0, y1, y2, y3, y4, y5, y6, y7;
    if (VAR == 0) {
      asm volatile(STEP_A(0, 0, "v[106:107]", "v108", "v109") STEP_A(1, 1, "v[108:109]", "v106", "v107") STEP_A(2, 2, "v[106:107]", "v108", "v109")
                   STEP_A(3, 3, "v[108:109]", "v106", "v107") STEP_A(4, 4, "v[106:107]", "v108", "v109") STEP_A(5, 5, "v[108:109]", "v106", "v107")
                   STEP_A(6, 6, "v[106:107]", "v108", "v109") STEP_A(7, 7, "v[108:109]", "v106", "v107")
                   : [y0] "=&v"(y0), [y1] "=&v"(y1), [y2] "=&v"(y2), [y3] "=&v"(y3), [y4] "=&v"(y4), [y5] "=&v"(y5), [y6] "=&v"(y6), [y7] "=&v"(y7),
                     "+{v[100:101]}"(ic), "+{v[108:109]}"(q)
                   : [x0] "v"(x[j]), [x1] "v"(x[j + 1]), [x2] "v"(x[j + 2]), [x3] "v"(x[j + 3]), [x4] "v"(x[j + 4]), [x5] "v"(x[j + 5]), [x6] "v"(x[j + 6]),
                     [x7] "v"(x[j + 7]), [a12] "v"(a12), [a23] "v"(a23)
                   : "v102", "v103", "v104", "v105", "v106", "v107", "v114", "v115");
    } else if (VAR == 1) {
      asm volatile(STEP_B(0, 0, "v[106:107]", "v106", "v107", "v108", "v109") STEP_B(1, 1, "v[108:109]", "v108", "v109", "v106", "v107")
                   STEP_B(2, 2, "v[106:107]", "v106", "v107", "v108", "v109") STEP_B(3, 3, "v[108:109]", "v108", "v109", "v106", "v107")
                   STEP_B(4, 4, "v[106:107]", "v106", "v107", "v108", "v109") STEP_B(5, 5, "v[108:109]", "v108", "v109", "v106", "v107")
                   STEP_B(6, 6, "v[106:107]", "v106", "v107", "v108", "v109") STEP_B(7, 7, "v[108:109]", "v108", "v109", "v106", "v107")
                   : [y0] "=&v"(y0), [y1] "=&v"(y1), [y2] "=&v"(y2), [y3] "=&v"(y3), [y4] "=&v"(y4), [y5] "=&v"(y5), [y6] "=&v"(y6), [y7] "=&v"(y7),
                     "+{v[100:101]}"(ic), "+{v[108:109]}"(q)
                   : [x0] "v"(x[j]), [x1] "v"(x[j + 1]), [x2] "v"(x[j + 2]), [x3] "v"(x[j + 3]), [x4] "v"(x[j + 4]), [x5] "v"(x[j + 5]), [x6] "v"(x[j + 6]),
                     [x7] "v"(x[j + 7]), [a12] "v"(a12), [a23] "v"(a23)
                   : "v102", "v103", "v104", "v105", "v106", "v107", "v114", "v115");
    } else if (VAR == 3) {
      asm volatile(STEP_D(0, 0, "v[106:107]", "v108", "v109") STEP_D(1, 1, "v[108:109]", "v106", "v107") STEP_D(2, 2, "v[106:107]", "v108", "v109")
                   STEP_D(3, 3, "v[108:109]", "v106", "v107") STEP_D(4, 4, "v[106:107]", "v108", "v109") STEP_D(5, 5, "v[108:109]", "v106", "v107")
                   STEP_D(6, 6, "v[106:107]", "v108", "v109") STEP_D(7, 7, "v[108:109]", "v106", "v107")
                   : [y0] "=&v"(y0), [y1] "=&v"(y1), [y2] "=&v"(y2), [y3] "=&v"(y3), [y4] "=&v"(y4), [y5] "=&v"(y5), [y6] "=&v"(y6), [y7] "=&v"(y7),
                     "+{v[100:101]}"(ic), "+{v[108:109]}"(q)
                   : [x0] "v"(x[j]), [x1] "v"(x[j + 1]), [x2] "v"(x[j + 2]), [x3] "v"(x[j + 3]), [x4] "v"(x[j + 4]), [x5] "v"(x[j + 5]), [x6] "v"(x[j + 6]),
                     [x7] "v"(x[j + 7]), [a12] "v"(a12), [a23] "v"(a23)
                   : "v102", "v103", "v104", "v105", "v106", "v107", "v114", "v115");
    } else {
      asm volatile(STEP_C(0, 0, "v[106:107]", "v108", "v109") STEP_C(1, 1, "v[108:109]", "v106", "v107") STEP_C(2, 2, "v[106:107]", "v108", "v109")
                   STEP_C(3, 3, "v[108:109]", "v106", "v107") STEP_C(4, 4, "v[106:107]", "v108", "v109") STEP_C(5, 5, "v[108:109]", "v106", "v107")
                   STEP_C(6, 6, "v[106:107]", "v108", "v109") STEP_C(7, 7, "v[108:109]", "v106", "v107")
                   : [y0] "=&v"(y0), [y1] "=&v"(y1), [y2] "=&v"(y2), [y3] "=&v"(y3), [y4] "=&v"(y4), [y5] "=&v"(y5), [y6] "=&v"(y6), [y7] "=&v"(y7),
                     "+{v[100:101]}"(ic), "+{v[108:109]}"(q)
                   : [x0] "v"(x[j]), [x1] "v"(x[j + 1]), [x2] "v"(x[j + 2]), [x3] "v"(x[j + 3]), [x4] "v"(x[j + 4]), [x5] "v"(x[j + 5]), [x6] "v"(x[j + 6]),
                     [x7] "v"(x[j + 7]), [a12] "v"(a12), [a23] "v"(a23)
                   : "v102", "v103", "v104", "v105", "v106", "v107", "v114", "v115");
    }
    if (j > 0) x[j - 1] = y0;
    x[j] = y1; x[j + 1] = y2; x[j + 2] = y3; x[j + 3] = y4; x[j + 4] = y5; x[j + 5] = y6; x[j + 6] = y7;
  }
  asm volatile("s_nop 0\n\tv_fma_f32 %[xout], 0, v108, v109" : [xout] "=v"(x[63]) : "{v[108:109]}"(q));
  ic1 = ic.x; ic2 = ic.y;
}

template <int VAR>
__global__ void __launch_bounds__(64) k(float* out, unsigned long long* ticks, int tiles) {
  // as in the pipeline kernel: the tile comes out of LDS rows and goes back into LDS rows; only the steps are timed.
  // The input is a bounded signal (two tiles, taken in turn), the coefficients those of a 1 kHz low-pass at 48 kHz, q = 0.7:
  // the filter state stays finite, and every tile's results are folded into a hash.
  __shared__ __attribute__((aligned(16))) float rows_in[2][64][68];
  __shared__ __attribute__((aligned(16))) float rows_out[64][68];
  typedef float V4 __attribute__((ext_vector_type(4)));
  float ic1 = 0.0f, ic2 = 0.0f;
  const f2 a12 = {0.91089f, 0.059664f}, a23 = {0.059664f, 0.0039080f};
  for (int i = threadIdx.x; i < 2 * 64 * 68; i += 64) (&rows_in[0][0][0])[i] = __sinf(0.37f * i) * 0.8f;
  __syncthreads();
  unsigned long long busy = 0;
  unsigned h = 2166136261u;
  for (int t = 0; t < tiles; ++t) {
    float x[64];
    const V4* in = reinterpret_cast<const V4*>(&rows_in[t & 1][threadIdx.x][0]);
#pragma unroll
    for (int j = 0; j < 16; ++j) { const V4 v = in[j]; x[4 * j] = v[0]; x[4 * j + 1] = v[1]; x[4 * j + 2] = v[2]; x[4 * j + 3] = v[3]; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    tile<VAR>(ic1, ic2, a12, a23, x);
    asm volatile("" ::: "memory");
    busy += __builtin_amdgcn_s_memtime() - t0;
    V4* o = reinterpret_cast<V4*>(&rows_out[threadIdx.x][0]);
#pragma unroll
    for (int j = 0; j < 16; ++j) { V4 v = {x[4 * j], x[4 * j + 1], x[4 * j + 2], x[4 * j + 3]}; o[j] = v; }
    __syncthreads();
    if (t < 64) {  // (the first tiles: the transient and the steady state)
      for (int j = 0; j < 64; ++j) h = (h ^ __builtin_bit_cast(unsigned, rows_out[threadIdx.x][j])) * 16777619u;
    }
    __syncthreads();
  }
  out[blockIdx.x * 64 + threadIdx.x] = __builtin_bit_cast(float, (h & 0x007FFFFFu) | 0x3F800000u) + 0.0f * (ic1 + ic2 == ic1 + ic2 ? 0.0f : 1.0f);
  if (threadIdx.x == 0 && blockIdx.x == 3) { ticks[0] = busy; ticks[1] = __builtin_bit_cast(unsigned, ic1); ticks[2] = __builtin_bit_cast(unsigned, ic2); }
}

template <int VAR>
void run(const char* name, float* d) {
  unsigned long long* ticks;
  (void)hipHostMalloc(&ticks, 64);
  const int tiles = 20000;
  k<VAR><<<256, 64>>>(d, ticks, 2000);
  (void)hipDeviceSynchronize();
  k<VAR><<<256, 64>>>(d, ticks, tiles);
  (void)hipDeviceSynchronize();
  static float host[256 * 64];
  (void)hipMemcpy(host, d, sizeof(host), hipMemcpyDeviceToHost);
  unsigned h = 2166136261u;
  for (int i = 0; i < 256 * 64; ++i) { unsigned b; __builtin_memcpy(&b, &host[i], 4); h = (h ^ b) * 16777619u; }
  float s1, s2; unsigned u1 = (unsigned)ticks[1], u2 = (unsigned)ticks[2];
  __builtin_memcpy(&s1, &u1, 4); __builtin_memcpy(&s2, &u2, 4);
  std::printf("%-72s %6.2f ticks per sample   (hash of the results %08x; final state %g %g)\n", name, (double)ticks[0] / (tiles * 64.0), h, s1, s2);
  (void)hipHostFree(ticks);
}

int main() {
  float* d;
  (void)hipMalloc(&d, 256 * 64 * 4);
  run<0>("A  shipped: 7 instructions + s_nop (packed state update)", d);
  run<1>("B  8 instructions, the state update as two scalar fmas", d);
  run<2>("C  the recurrence alone, output as a move (not a usable step)", d);
  run<3>("D  A without its s_nop: does the hardware interlock, and what does it cost?", d);
  run<4>("E  six instructions: (v1, v2) in pairs of the compiler's choice, output = v2, a finiteness test per run of eight", d);
  run<0>("A  again", d);
  return 0;
}
