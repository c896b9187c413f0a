// Time per sample of the SVF recurrence (svf.rs:272-278) for a wavefront alone on its SIMD, in several encodings.
// The arithmetic is the same in all of them (15 roundings for the full step, 10 for the recurrence alone).
//   hipcc --offload-arch=gfx950 -O3 -o svf_chain svf_chain.hip && ./svf_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

// registers: v[100:101] (ic1, ic2)  v[102:103] P1  v[104:105] P2  v[106:107] (v1, v2)  v[108:109] Q  v112 sum  v114 v3  v116.. scratch
#define FULL10(X)                                                              \
  "v_pk_mul_f32 v[102:103], %[a12], v[100:101] op_sel_hi:[1,0]\n\t"            \
  "v_sub_f32 v114, %[" #X "], v101\n\t"                                        \
  "v_add_f32 v112, v112, v108\n\t"                                             \
  "v_add_f32 v103, v101, v103\n\t"                                             \
  "v_pk_mul_f32 v[104:105], %[a23], v[114:115] op_sel_hi:[1,0]\n\t"            \
  "v_add_f32 v116, v112, v109\n\t"                                             \
  "v_pk_add_f32 v[106:107], v[102:103], v[104:105]\n\t"                        \
  "v_mul_f32 v112, %[m0], %[" #X "]\n\t"                                       \
  "v_pk_fma_f32 v[100:101], v[106:107], 2.0, v[100:101] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t" \
  "v_pk_mul_f32 v[108:109], %[m12], v[106:107]\n\t"
#define CORE6_NOP(X)                                                           \
  "v_sub_f32 v114, %[" #X "], v101\n\t"                                        \
  "v_pk_mul_f32 v[102:103], %[a12], v[100:101] op_sel_hi:[1,0]\n\t"            \
  "v_pk_mul_f32 v[104:105], %[a23], v[114:115] op_sel_hi:[1,0]\n\t"            \
  "v_add_f32 v103, v101, v103\n\t"                                             \
  "v_pk_add_f32 v[106:107], v[102:103], v[104:105]\n\t"                        \
  "s_nop 0\n\t"                                                                \
  "v_pk_fma_f32 v[100:101], v[106:107], 2.0, v[100:101] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t" \
  "s_nop 0\n\t"
// the two wait states filled with unrelated work instead of s_nop
#define CORE6_FILL(X)                                                          \
  "v_sub_f32 v114, %[" #X "], v101\n\t"                                        \
  "v_pk_mul_f32 v[102:103], %[a12], v[100:101] op_sel_hi:[1,0]\n\t"            \
  "v_pk_mul_f32 v[104:105], %[a23], v[114:115] op_sel_hi:[1,0]\n\t"            \
  "v_add_f32 v103, v101, v103\n\t"                                             \
  "v_pk_add_f32 v[106:107], v[102:103], v[104:105]\n\t"                        \
  "v_mul_f32 v116, %[m0], %[" #X "]\n\t"                                       \
  "v_pk_fma_f32 v[100:101], v[106:107], 2.0, v[100:101] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t" \
  "v_mul_f32 v117, %[m0], %[" #X "]\n\t"
// scalar arithmetic on the v3 -> v2 -> ic2 path, packed only off that path; state update with one packed fma
#define CORE8_SC(X)                                                            \
  "v_sub_f32 v114, %[" #X "], v101\n\t"                                        \
  "v_pk_mul_f32 v[102:103], %[a12], v[100:101] op_sel_hi:[1,0]\n\t"            \
  "v_mul_f32 v105, %[a3], v114\n\t"                                            \
  "v_mul_f32 v104, %[a2], v114\n\t"                                            \
  "v_add_f32 v103, v101, v103\n\t"                                             \
  "v_add_f32 v106, v102, v104\n\t"                                             \
  "v_add_f32 v107, v103, v105\n\t"                                             \
  "s_nop 0\n\t"                                                                \
  "v_pk_fma_f32 v[100:101], v[106:107], 2.0, v[100:101] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t" \
  "s_nop 0\n\t"
// the same with two scalar fmas for the state
#define CORE9_SC(X)                                                            \
  "v_sub_f32 v114, %[" #X "], v101\n\t"                                        \
  "v_pk_mul_f32 v[102:103], %[a12], v[100:101] op_sel_hi:[1,0]\n\t"            \
  "v_mul_f32 v105, %[a3], v114\n\t"                                            \
  "v_mul_f32 v104, %[a2], v114\n\t"                                            \
  "v_add_f32 v103, v101, v103\n\t"                                             \
  "v_add_f32 v107, v103, v105\n\t"                                             \
  "v_add_f32 v106, v102, v104\n\t"                                             \
  "v_fma_f32 v101, v107, 2.0, -v101\n\t"                                       \
  "v_fma_f32 v100, v106, 2.0, -v100\n\t"
// everything scalar
#define CORE10_SC(X)                                                           \
  "v_sub_f32 v114, %[" #X "], v101\n\t"                                        \
  "v_mul_f32 v103, %[a2], v100\n\t"                                            \
  "v_mul_f32 v105, %[a3], v114\n\t"                                            \
  "v_mul_f32 v102, %[a1], v100\n\t"                                            \
  "v_add_f32 v103, v101, v103\n\t"                                             \
  "v_mul_f32 v104, %[a2], v114\n\t"                                            \
  "v_add_f32 v107, v103, v105\n\t"                                             \
  "v_add_f32 v106, v102, v104\n\t"                                             \
  "v_fma_f32 v101, v107, 2.0, -v101\n\t"                                       \
  "v_fma_f32 v100, v106, 2.0, -v100\n\t"
// only the chain itself: sub, mul, add, fma on the ic2 path (what no encoding can go below)
#define CHAIN4(X)                                                              \
  "v_sub_f32 v114, %[" #X "], v101\n\t"                                        \
  "v_mul_f32 v105, %[a3], v114\n\t"                                            \
  "v_add_f32 v107, v103, v105\n\t"                                             \
  "v_fma_f32 v101, v107, 2.0, -v101\n\t"
#define CHAIN4_PK(X)                                                           \
  "v_sub_f32 v114, %[" #X "], v101\n\t"                                        \
  "v_pk_mul_f32 v[104:105], %[a23], v[114:115] op_sel_hi:[1,0]\n\t"            \
  "s_nop 0\n\t"                                                                \
  "v_pk_add_f32 v[106:107], v[102:103], v[104:105]\n\t"                        \
  "s_nop 0\n\t"                                                                \
  "v_pk_fma_f32 v[100:101], v[106:107], 2.0, v[100:101] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t" \
  "s_nop 0\n\t"

// round 2: m0*x formed outside (two samples to a packed multiply, see PK_M0X), nine instructions per sample inside
#define FULL9_A(X)                                                             \
  "v_pk_mul_f32 v[102:103], %[a12], v[100:101] op_sel_hi:[1,0]\n\t"            \
  "v_sub_f32 v114, %[" #X "], v101\n\t"                                        \
  "v_add_f32 v103, v101, v103\n\t"                                             \
  "v_pk_mul_f32 v[104:105], %[a23], v[114:115] op_sel_hi:[1,0]\n\t"            \
  "v_add_f32 v112, %[m0], v108\n\t"                                            \
  "v_pk_add_f32 v[106:107], v[102:103], v[104:105]\n\t"                        \
  "v_add_f32 v116, v112, v109\n\t"                                             \
  "v_pk_fma_f32 v[100:101], v[106:107], 2.0, v[100:101] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t" \
  "v_pk_mul_f32 v[108:109], %[m12], v[106:107]\n\t"
// the same with the packed results read at distance 3 where possible, one s_nop in front of the state update
#define FULL9_D(X)                                                             \
  "v_pk_mul_f32 v[102:103], %[a12], v[100:101] op_sel_hi:[1,0]\n\t"            \
  "v_sub_f32 v114, %[" #X "], v101\n\t"                                        \
  "v_add_f32 v112, %[m0], v108\n\t"                                            \
  "v_add_f32 v103, v101, v103\n\t"                                             \
  "v_pk_mul_f32 v[104:105], %[a23], v[114:115] op_sel_hi:[1,0]\n\t"            \
  "v_add_f32 v116, v112, v109\n\t"                                             \
  "v_pk_add_f32 v[106:107], v[102:103], v[104:105]\n\t"                        \
  "s_nop 0\n\t"                                                                \
  "v_pk_fma_f32 v[100:101], v[106:107], 2.0, v[100:101] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t" \
  "v_pk_mul_f32 v[108:109], %[m12], v[106:107]\n\t"
// four packed multiplies per eight samples: what forming m0*x outside costs
#define PK_M0X                                                                  \
  "v_pk_mul_f32 v[118:119], %[a12], %[a23]\n\t" "v_pk_mul_f32 v[120:121], %[a12], %[a23]\n\t"                          \
  "v_pk_mul_f32 v[122:123], %[a12], %[a23]\n\t" "v_pk_mul_f32 v[124:125], %[a12], %[a23]\n\t"
#define EIGHT(S) S(x0) S(x1) S(x2) S(x3) S(x4) S(x5) S(x6) S(x7)
#define RUN(S)                                                                                                          \
  asm volatile(EIGHT(S) : "+{v[100:101]}"(ic), "+{v[108:109]}"(q), "+{v112}"(o)                                         \
               : [x0] "v"(x[0]), [x1] "v"(x[1]), [x2] "v"(x[2]), [x3] "v"(x[3]), [x4] "v"(x[4]), [x5] "v"(x[5]),        \
                 [x6] "v"(x[6]), [x7] "v"(x[7]), [a12] "v"(a12), [a23] "v"(a23), [m12] "v"(m12), [m0] "v"(m0),           \
                 [a1] "v"(a12.x), [a2] "v"(a12.y), [a3] "v"(a23.y)                                                       \
               : "v102", "v103", "v104", "v105", "v106", "v107", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121",      \
                 "v122", "v123", "v124", "v125")
#define EIGHT_M(S) PK_M0X S(x0) S(x1) S(x2) S(x3) S(x4) S(x5) S(x6) S(x7)
#define RUN_M(S)                                                                                                        \
  asm volatile(EIGHT_M(S) : "+{v[100:101]}"(ic), "+{v[108:109]}"(q), "+{v112}"(o)                                       \
               : [x0] "v"(x[0]), [x1] "v"(x[1]), [x2] "v"(x[2]), [x3] "v"(x[3]), [x4] "v"(x[4]), [x5] "v"(x[5]),        \
                 [x6] "v"(x[6]), [x7] "v"(x[7]), [a12] "v"(a12), [a23] "v"(a23), [m12] "v"(m12), [m0] "v"(m0),           \
                 [a1] "v"(a12.x), [a2] "v"(a12.y), [a3] "v"(a23.y)                                                       \
               : "v102", "v103", "v104", "v105", "v106", "v107", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121",      \
                 "v122", "v123", "v124", "v125")

template <int MODE>
__global__ void k(float* out, unsigned long long* ticks, int iters) {
  f2 ic = {0.0f, 0.0f}, q = {0.0f, 0.0f};
  const f2 a12 = {0.98f, 0.07f}, a23 = {0.07f, 0.005f}, m12 = {0.0f, 1.0f};
  const float m0 = 0.0f;
  float o = 0.0f;
  float x[8];
  for (int j = 0; j < 8; ++j) x[j] = __sinf(0.1f * (threadIdx.x + j));
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) RUN(FULL10);
    else if (MODE == 1) RUN(CORE6_NOP);
    else if (MODE == 2) RUN(CORE6_FILL);
    else if (MODE == 3) RUN(CORE8_SC);
    else if (MODE == 4) RUN(CORE9_SC);
    else if (MODE == 5) RUN(CORE10_SC);
    else if (MODE == 6) RUN(CHAIN4);
    else if (MODE == 7) RUN(CHAIN4_PK);
    else if (MODE == 8) RUN(FULL9_A);
    else if (MODE == 9) RUN_M(FULL9_A);
    else if (MODE == 10) RUN(FULL9_D);
    else if (MODE == 11) RUN_M(FULL9_D);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = ic.x + ic.y + q.x + q.y + o;
  if (threadIdx.x == 0 && blockIdx.x == 7) ticks[0] = t1 - t0;
}

template <int MODE>
void run(const char* name, float* d, int waves) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  unsigned long long* ticks;
  (void)hipHostMalloc(&ticks, 8);
  k<MODE><<<256, 64 * waves>>>(d, ticks, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<256, 64 * waves>>>(d, ticks, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  // s_memtime ticks per sample of one wave: with the ns figure, the tick rate (a fixed 100 MHz reference on gfx950)
  std::printf("%-58s waves/CU %d: %6.2f ns per sample, %7.3f s_memtime ticks per sample\n", name, waves, ms * 1e6 / (iters * 8.0),
              (double)ticks[0] / (iters * 8.0));
  (void)hipHostFree(ticks);
}

int main() {
  float* d;
  (void)hipMalloc(&d, 256 * 1024 * 4);
  for (int waves : {1, 2, 4, 8}) {
    run<0>("full step, 10 instr (5 packed), as shipped", d, waves);
    run<8>("full step, 9 instr inside (m0*x given)", d, waves);
    run<9>("full step, 9 instr + 4 packed m0*x per 8 samples (shipped r2)", d, waves);
    run<10>("full step, 9 instr + s_nop, packed results at distance 3", d, waves);
    run<11>("the same + 4 packed m0*x per 8 samples", d, waves);
    run<1>("recurrence only, 6 instr (4 packed) + 2 s_nop", d, waves);
    run<2>("recurrence only, 6 instr, wait states filled with work", d, waves);
    run<3>("recurrence, scalar on the ic2 path, 8 instr + 2 s_nop", d, waves);
    run<4>("recurrence, scalar on the ic2 path, scalar fmas, 9 instr", d, waves);
    run<5>("recurrence, all scalar, 10 instr", d, waves);
    run<6>("the ic2 dependency chain alone, scalar (4 instr)", d, waves);
    run<7>("the ic2 dependency chain alone, packed (4 instr + 3 s_nop)", d, waves);
  }
  return 0;
}
