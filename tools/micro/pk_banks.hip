// Do the source registers of a packed f32 instruction matter?  VGPRs sit in four banks (register number mod 4); a 64-bit
// operand takes two neighbouring ones.  Eight independent chains of  v_pk_mul_f32 d, d, s  /  v_pk_fma_f32 d, d, s, t  with
// the operands placed by hand: every operand pair in a bank pair of its own, two of them in the same bank pair, all three.
// 1, 2 and 4 wavefronts per SIMD.  (Round 4: the 262 144-voice bank runs with its VALUs 100 % busy at 4.7 cycles per
// instruction; tools/micro/valu_issue.hip has packed instructions at 3.7 .. 7 cycles depending on the test.)
//   hipcc --offload-arch=gfx950 -O3 -o pk_banks pk_banks.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHAIN8(OP, S1, S2)                                                     \
  OP(20, S1, S2) OP(24, S1, S2) OP(28, S1, S2) OP(32, S1, S2) OP(36, S1, S2) OP(40, S1, S2) OP(44, S1, S2) OP(48, S1, S2)
// d = v[D:D+1] (D = 20, 24, ..: bank pair 0,1)
#define PK_MUL(D, S1, S2) "v_pk_mul_f32 v[" #D ":" #D "+1], v[" #D ":" #D "+1], v[" #S1 ":" #S1 "+1]\n\t"
#define PK_FMA(D, S1, S2) "v_pk_fma_f32 v[" #D ":" #D "+1], v[" #D ":" #D "+1], v[" #S1 ":" #S1 "+1], v[" #S2 ":" #S2 "+1]\n\t"
#define SC_MUL(D, S1, S2) "v_mul_f32 v" #D ", v" #D ", v" #S1 "\n\t"
#define SC_FMA(D, S1, S2) "v_fma_f32 v" #D ", v" #D ", v" #S1 ", v" #S2 "\n\t"

#define CLOBBERS "v20","v21","v24","v25","v28","v29","v32","v33","v36","v37","v40","v41","v44","v45","v48","v49","v60","v61","v62","v63","v64","v65","v66","v67"

template <int MODE>
__global__ void __launch_bounds__(1024) k(float* out, int iters) {
  float acc = 0.f;
  asm volatile(
      "v_mov_b32 v60, 0x3f7fff00\n\tv_mov_b32 v61, 0x3f7ffe00\n\tv_mov_b32 v62, 0x3f7ffd00\n\tv_mov_b32 v63, 0x3f7ffc00\n\t"
      "v_mov_b32 v64, 0\n\tv_mov_b32 v65, 0\n\tv_mov_b32 v66, 0\n\tv_mov_b32 v67, 0\n\t"
      "v_mov_b32 v20, 1.0\n\tv_mov_b32 v21, 1.0\n\tv_mov_b32 v24, 1.0\n\tv_mov_b32 v25, 1.0\n\tv_mov_b32 v28, 1.0\n\tv_mov_b32 v29, 1.0\n\t"
      "v_mov_b32 v32, 1.0\n\tv_mov_b32 v33, 1.0\n\tv_mov_b32 v36, 1.0\n\tv_mov_b32 v37, 1.0\n\tv_mov_b32 v40, 1.0\n\tv_mov_b32 v41, 1.0\n\t"
      "v_mov_b32 v44, 1.0\n\tv_mov_b32 v45, 1.0\n\tv_mov_b32 v48, 1.0\n\tv_mov_b32 v49, 1.0\n\t" ::: CLOBBERS);
  for (int i = 0; i < iters; ++i) {
    // sources: v[60:61] = banks 0,1 (the destinations' own pair), v[62:63] = banks 2,3, v[64:65] = banks 0,1, v[66:67] = banks 2,3
    if (MODE == 0) asm volatile(CHAIN8(PK_MUL, 62, 0) CHAIN8(PK_MUL, 62, 0) ::: CLOBBERS);        // d(0,1) * s(2,3)
    else if (MODE == 1) asm volatile(CHAIN8(PK_MUL, 60, 0) CHAIN8(PK_MUL, 60, 0) ::: CLOBBERS);   // d(0,1) * s(0,1): same banks
    else if (MODE == 2) asm volatile(CHAIN8(PK_FMA, 62, 66) CHAIN8(PK_FMA, 62, 66) ::: CLOBBERS); // d(0,1), s(2,3), t(2,3)
    else if (MODE == 3) asm volatile(CHAIN8(PK_FMA, 60, 64) CHAIN8(PK_FMA, 60, 64) ::: CLOBBERS); // all three in banks 0,1
    else if (MODE == 4) asm volatile(CHAIN8(PK_FMA, 62, 64) CHAIN8(PK_FMA, 62, 64) ::: CLOBBERS); // d(0,1), s(2,3), t(0,1)
    else if (MODE == 5) asm volatile(CHAIN8(SC_MUL, 62, 0) CHAIN8(SC_MUL, 62, 0) ::: CLOBBERS);   // scalar: d bank 0, s bank 2
    else if (MODE == 6) asm volatile(CHAIN8(SC_MUL, 60, 0) CHAIN8(SC_MUL, 60, 0) ::: CLOBBERS);   // scalar: d bank 0, s bank 0
    else if (MODE == 7) asm volatile(CHAIN8(SC_FMA, 61, 62) CHAIN8(SC_FMA, 61, 62) ::: CLOBBERS); // scalar fma: banks 0, 1, 2
    else if (MODE == 8) asm volatile(CHAIN8(SC_FMA, 60, 64) CHAIN8(SC_FMA, 60, 64) ::: CLOBBERS); // scalar fma: banks 0, 0, 0
  }
  asm volatile("v_add_f32 %0, v20, v25" : "=v"(acc) :: CLOBBERS);
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int MODE>
void run(const char* name, float* d) {
  std::printf("%-52s", name);
  for (int waves : {4, 8, 16}) {
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
    // per SIMD: waves / 4 wavefronts issue iters * 16 instructions each
    std::printf("  %6.3f", ms * 1e6 / (iters * 16.0) / (waves / 4));
  }
  std::printf("\n");
}

int main() {
  float* d;
  (void)hipMalloc(&d, 256 * 1024 * 4);
  std::printf("ns per instruction per SIMD at 1, 2, 4 wavefronts per SIMD (eight independent chains per wavefront)\n");
  run<0>("v_pk_mul_f32  d(banks 0,1) * s(2,3)", d);
  run<1>("v_pk_mul_f32  d(0,1) * s(0,1)", d);
  run<2>("v_pk_fma_f32  d(0,1), s(2,3), t(2,3)", d);
  run<4>("v_pk_fma_f32  d(0,1), s(2,3), t(0,1)", d);
  run<3>("v_pk_fma_f32  d(0,1), s(0,1), t(0,1)", d);
  run<5>("v_mul_f32     d(0) * s(2)", d);
  run<6>("v_mul_f32     d(0) * s(0)", d);
  run<7>("v_fma_f32     d(0), s(1), t(2)", d);
  run<8>("v_fma_f32     d(0), s(0), t(0)", d);
  return 0;
}
