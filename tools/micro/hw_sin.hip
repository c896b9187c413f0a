// How far is the hardware sine (v_sin_f32: sin(2*pi*x), x in revolutions) from sin(fl(x * TAU)) as the reference computes it
// (osc.rs: (phase * TAU).sin() in f32, glibc sinf = the double value rounded)?  And the device library's sinf for comparison.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/hw_sin tools/micro/hw_sin.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__global__ void k(const float* p, float* hw, float* lib, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  hw[i] = __builtin_amdgcn_sinf(p[i]);
  lib[i] = __ocml_sin_f32(p[i] * 6.28318530717958647692f);
}

int main() {
  const int n = 1 << 24;
  std::vector<float> p(n), hw(n), lib(n);
  // every 2^-24 step of [0, 1) (all the phases an f32 accumulator in [0.5, 1) can hold, and more below), then [1, 2) coarser
  for (int i = 0; i < n; ++i) p[i] = (float)i / (float)n;
  float *dp, *dh, *dl;
  (void)hipMalloc(&dp, n * 4); (void)hipMalloc(&dh, n * 4); (void)hipMalloc(&dl, n * 4);
  for (int pass = 0; pass < 4; ++pass) {  // [0, 1), [1, 2), [-1, 0), [-2, -1): every f32 of each range that an f32 step of 2^-24 reaches
    if (pass == 1) for (int i = 0; i < n; ++i) p[i] = 1.0f + (float)i / (float)n;
    if (pass == 2) for (int i = 0; i < n; ++i) p[i] = -(float)(i + 1) / (float)n;
    if (pass == 3) for (int i = 0; i < n; ++i) p[i] = -1.0f - (float)i / (float)n;
    (void)hipMemcpy(dp, p.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dp, dh, dl, n);
    (void)hipMemcpy(hw.data(), dh, n * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(lib.data(), dl, n * 4, hipMemcpyDeviceToHost);
    double e_hw = 0, e_lib = 0, e_hw_true = 0;
    int at = 0;
    for (int i = 0; i < n; ++i) {
      const float arg = p[i] * 6.28318530717958647692f;            // the reference's f32 argument
      const double ref = (double)(float)std::sin((double)arg);      // glibc sinf(arg): correctly rounded in practice
      const double d = std::fabs((double)hw[i] - ref);
      if (d > e_hw) { e_hw = d; at = i; }
      e_lib = std::fmax(e_lib, std::fabs((double)lib[i] - ref));
      e_hw_true = std::fmax(e_hw_true, std::fabs((double)hw[i] - std::sin(6.283185307179586476925 * (double)p[i])));
    }
    std::printf("pass %d: max |v_sin_f32(p) - sinf(fl(p*TAU))| = %.3e (at p = %.9g: %.9g)   device sinf: %.3e   v_sin_f32 against sin(2 pi p) in double: %.3e\n",
                pass, e_hw, (double)p[at], (double)hw[at], e_lib, e_hw_true);
  }
  return 0;
}
