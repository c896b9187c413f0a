// What the register butterfly's exchange instructions cost a wavefront: v_permlane32_swap, v_permlane16_swap and a DPP add,
// 16 independent ones in a row, against v_add_f32; one wavefront per SIMD and two.
//   hipcc --offload-arch=gfx950 -O3 -o swap_cost swap_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void __launch_bounds__(512) k(float* out, int iters) {
  float a[16];
  for (int j = 0; j < 16; ++j) a[j] = threadIdx.x * 0.001f + j;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; u += 2) {
      if (MODE == 0) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[u]), "+v"(a[u + 1]));
      else if (MODE == 1) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[u]), "+v"(a[u + 1]));
      else if (MODE == 2) asm volatile("v_add_f32 %0, %0, %1 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(a[u]) : "v"(a[u + 1]));
      else if (MODE == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[u]) : "v"(a[u + 1]));
      else if (MODE == 4) asm volatile("v_permlane32_swap_b32 %0, %1\n\tv_add_f32 %0, %0, %1" : "+v"(a[u]), "+v"(a[u + 1]));
    }
  }
  float s = 0;
  for (int j = 0; j < 16; ++j) s += a[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, float* d, int per) {
  std::printf("%-44s", name);
  for (int waves : {4, 8}) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;
    k<MODE><<<256, 64 * waves>>>(d, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<MODE><<<256, 64 * waves>>>(d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::printf("  %7.3f", ms * 1e6 / (iters * 8.0 * per) / (waves / 4));
  }
  std::printf("\n");
}
int main() {
  float* d;
  (void)hipMalloc(&d, 256 * 1024 * 4);
  std::printf("ns per instruction per SIMD, one and two wavefronts per SIMD (eight independent instructions per round)\n");
  run<0>("v_permlane32_swap", d, 1);
  run<1>("v_permlane16_swap", d, 1);
  run<2>("v_add_f32 ... row_mirror (DPP)", d, 1);
  run<3>("v_add_f32", d, 1);
  run<4>("v_permlane32_swap + dependent v_add_f32 (per instr)", d, 2);
  return 0;
}
