// Does the clock a wavefront runs at depend on how busy the OTHER three SIMDs of its CU are?  (gfx950 DVFS)
// 256 workgroups x 4 wavefronts (one per SIMD).  Wave 0 of each runs the 10-instruction SVF step `iters` x 8 times;
// waves 1-3 run the same loop for a fraction of that and leave.  Reported: wall ns per sample of wave 0 and its
// shader-clock cycles per sample (s_memtime), hence the clock.
//   hipcc --offload-arch=gfx950 -O3 -o clock_share clock_share.hip && ./clock_share
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
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
#define EIGHT(S) S(x0) S(x1) S(x2) S(x3) S(x4) S(x5) S(x6) S(x7)

__global__ void k(float* out, unsigned long long* ticks, int iters, int others_iters, int lanes) {
  f2 ic = {0.0f, 0.0f}, q = {0.0f, 0.0f};
  const f2 a12 = {0.98f, 0.07f}, a23 = {0.07f, 0.005f}, m12 = {0.0f, 1.0f};
  const float m0 = 0.0f;
  float o = 0.0f;
  float x[8];
  for (int j = 0; j < 8; ++j) x[j] = __sinf(0.1f * (threadIdx.x + j));
  const int wave = threadIdx.x >> 6;
  const int n = wave == 0 ? iters : others_iters;
  if ((int)(threadIdx.x & 63) >= lanes && wave != 0) return;  // the other waves with fewer lanes
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i)
    asm volatile(EIGHT(FULL10) : "+{v[100:101]}"(ic), "+{v[108:109]}"(q), "+{v112}"(o)
                 : [x0] "v"(x[0]), [x1] "v"(x[1]), [x2] "v"(x[2]), [x3] "v"(x[3]), [x4] "v"(x[4]), [x5] "v"(x[5]),
                   [x6] "v"(x[6]), [x7] "v"(x[7]), [a12] "v"(a12), [a23] "v"(a23), [m12] "v"(m12), [m0] "v"(m0)
                 : "v102", "v103", "v104", "v105", "v106", "v107", "v114", "v115", "v116", "v117");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = ic.x + ic.y + q.x + q.y + o;
  if (threadIdx.x == 0 && blockIdx.x == 7) ticks[0] = t1 - t0;
}

int main() {
  float* d;
  unsigned long long* ticks;
  (void)hipMalloc(&d, 256 * 256 * 4);
  (void)hipHostMalloc(&ticks, 8);
  const int iters = 400000;  // ~80 ms per run: long against any clock-control loop
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  struct { const char* name; int others; int lanes; } cases[] = {
      {"other three SIMDs idle", 0, 64}, {"other three busy 25 % of the time", iters / 4, 64},
      {"other three busy 50 %", iters / 2, 64}, {"other three busy 75 %", 3 * iters / 4, 64},
      {"other three busy 100 %", iters, 64}, {"other three busy 100 %, 32 lanes each", iters, 32},
      {"other three busy 100 %, 8 lanes each", iters, 8}};
  for (auto& c : cases) {
    k<<<256, 256>>>(d, ticks, 1000, 1000, 64);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<<<256, 256>>>(d, ticks, iters, c.others, c.lanes);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / (iters * 8.0), cyc = (double)ticks[0] / (iters * 8.0);
    std::printf("%-42s wave 0: %6.2f ns per sample, %6.2f cycles per sample -> %.2f GHz average\n", c.name, ns, cyc, cyc / ns);
  }
  return 0;
}
