// kernels_pipe.hip -- gfx950 instantiations of the wave-specialised (pipelined) voice-bank kernel (voice_pipe.hpp).
// Built with -ffp-contract=off, as every kernel of the library.
#include <cstring>

#include "kernel_registry.hpp"
#include "voice_pipe.hpp"

namespace knh {
using namespace knh_dev;

// FORM: PIPE_MIXER = 32-sample tiles (16 for f64) and a mixer wavefront; PIPE_FOLD / PIPE_INPLACE = 64-sample tiles (32 for
// f64) with the fold done by the last stage group / by a mixer wavefront behind a last group that works in place
template <typename F, bool FMA, int FORM, typename... Gs>
static hipError_t launch_pipe(const VoiceKernelArgs<F>& args, unsigned n_wavefronts, hipStream_t stream) {
  if (n_wavefronts == 0) return hipSuccess;
  constexpr int T = FORM != PIPE_MIXER ? PipeTile<F>::big : PipeTile<F>::value;
  hipLaunchKernelGGL((voice_pipe_kernel<F, FMA, T, FORM, 1, Gs...>), dim3(n_wavefronts), dim3(PipeWaves<T, FORM, Gs...>::value * 64), 0, stream, args);
  return hipGetLastError();
}
// Two 64-voice groups per workgroup, short tiles (32 samples, f64: 16), the last stage group in place and a mixer wavefront
// per group: for banks of more voice groups than the chip has CUs (voice_pipe.hpp, GPW)
template <typename F, bool FMA, typename... Gs>
static hipError_t launch_pipe_pair(const VoiceKernelArgs<F>& args, unsigned n_wavefronts, hipStream_t stream) {
  if (n_wavefronts == 0) return hipSuccess;
  constexpr int T = PipeTile<F>::value;
  hipLaunchKernelGGL((voice_pipe_kernel<F, FMA, T, PIPE_INPLACE, 2, Gs...>), dim3((n_wavefronts + 1u) / 2u),
                     dim3(2 * PipeWaves<T, PIPE_INPLACE, Gs...>::value * 64), 0, stream, args);
  return hipGetLastError();
}
// 64-sample tiles (32 for f64) WITH a mixer wavefront: for pipelines with a Fan group (the fold costs the same per
// tile whatever the tile length, and a Fan group's windows are whole runs of eight samples)
template <typename F, bool FMA, typename... Gs>
static hipError_t launch_pipe_wide_tile(const VoiceKernelArgs<F>& args, unsigned n_wavefronts, hipStream_t stream) {
  if (n_wavefronts == 0) return hipSuccess;
  constexpr int T = PipeTile<F>::big;
  hipLaunchKernelGGL((voice_pipe_kernel<F, FMA, T, PIPE_MIXER, 1, Gs...>), dim3(n_wavefronts), dim3(PipeWaves<T, PIPE_MIXER, Gs...>::value * 64), 0, stream, args);
  return hipGetLastError();
}
#define KNH_PIPE_FAN(sig, n, ...)                                                                                       \
  {sig, n, PIPE_MIXER, 1, 1, {launch_pipe_wide_tile<float, false, __VA_ARGS__>, launch_pipe_wide_tile<float, true, __VA_ARGS__>}, \
   {launch_pipe_wide_tile<double, false, __VA_ARGS__>, launch_pipe_wide_tile<double, true, __VA_ARGS__>}}
#define KNH_PIPE_AS(sig, n, form, ...)                                                               \
  {sig, n, form, 1, form != PIPE_MIXER ? 1 : 0, {launch_pipe<float, false, form, __VA_ARGS__>, launch_pipe<float, true, form, __VA_ARGS__>}, \
   {launch_pipe<double, false, form, __VA_ARGS__>, launch_pipe<double, true, form, __VA_ARGS__>}}
#define KNH_PIPE_PAIR_AS(sig, n, ...)                                                                \
  {sig, n, PIPE_INPLACE, 2, 0, {launch_pipe_pair<float, false, __VA_ARGS__>, launch_pipe_pair<float, true, __VA_ARGS__>}, \
   {launch_pipe_pair<double, false, __VA_ARGS__>, launch_pipe_pair<double, true, __VA_ARGS__>}},
// This file is compiled once per form (KNH_PIPE_PART = PIPE_MIXER, PIPE_FOLD, PIPE_INPLACE: build.py), each time with the
// table of that form's kernels; find_pipe() lives in the PIPE_MIXER part and looks through all three.
#ifndef KNH_PIPE_PART
#error "KNH_PIPE_PART: 0, 1 or 2 (knaster_amd/build.py)"
#endif
#if KNH_PIPE_PART == 0
#define KNH_PIPE(sig, n, ...) KNH_PIPE_AS(sig, n, PIPE_MIXER, __VA_ARGS__),
#define KNH_PIPE_BIG(sig, n, ...)
#define KNH_PIPE_PAIR(sig, n, ...)
#define KNH_PIPE_FAN_(sig, n, ...) KNH_PIPE_FAN(sig, n, __VA_ARGS__),
#define KNH_PIPE_TABLE pipes_mixer
#elif KNH_PIPE_PART == 1
#define KNH_PIPE(sig, n, ...)
#define KNH_PIPE_BIG(sig, n, ...) KNH_PIPE_AS(sig, n, PIPE_FOLD, __VA_ARGS__),
#define KNH_PIPE_PAIR(sig, n, ...) KNH_PIPE_PAIR_AS(sig, n, __VA_ARGS__)
#define KNH_PIPE_FAN_(sig, n, ...)
#define KNH_PIPE_TABLE pipes_fold
#else
#define KNH_PIPE(sig, n, ...)
#define KNH_PIPE_BIG(sig, n, ...) KNH_PIPE_AS(sig, n, PIPE_INPLACE, __VA_ARGS__),
#define KNH_PIPE_PAIR(sig, n, ...)
#define KNH_PIPE_FAN_(sig, n, ...)
#define KNH_PIPE_TABLE pipes_inplace
#endif

typedef Group<SinWt, MulVal> G_Wm;
typedef Group<SinWt> G_W;
typedef Group<SinNum> G_N;
typedef Group<SinPhase> G_Np;                 // SinNumeric's phase accumulator ...
typedef Fan<8, SinMap, MulVal> F_Nm;          // ... and its sin (with the gain), over eight wavefronts: C2
typedef Fan<8, SinMap> F_N;
typedef Group<Svf> G_S;
typedef Group<MulAsr> G_A;
typedef Group<MulAr> G_E;
typedef Group<MulAsr, MulVal> G_Am;
typedef Group<MulVal> G_m;
typedef Group<SinWt, MulVal, AddVal> G_Wma;
typedef Group<SinWtAr, MulVal> G_Rm;
typedef Group<SampleDelay, MulAsr> G_DA;
typedef Group<PolyBlepOsc, MulVal> G_Bm;
typedef Group<MulAr, Pan2> G_E2;
typedef Group<MulAsr, Pan2> G_A2;

// (host-only tables: a namespace-scope constant with external linkage would be emitted for the device too)
const PipeEntry* pipes_mixer(int* n);
const PipeEntry* pipes_fold(int* n);
const PipeEntry* pipes_inplace(int* n);
static const PipeEntry kTable[] = {
    KNH_PIPE_BIG("WmSA", 3, G_Wm, G_S, G_A)   // C3/C4: oscillator | filter | envelope (+ fold), 64-sample tiles
    KNH_PIPE("WmSA", 3, G_Wm, G_S, G_A)       // the same with 32-sample tiles and a mixer wavefront (KNH_PIPE_BIG=0)
    KNH_PIPE_PAIR("WmSA", 3, G_Wm, G_S, G_A)  // two voice groups per workgroup (banks of more than 256 groups); built with the PIPE_FOLD part
    KNH_PIPE_PAIR("WSAm", 3, G_W, G_S, G_Am)
    KNH_PIPE_BIG("WSAm", 3, G_W, G_S, G_Am)
    KNH_PIPE("WSAm", 3, G_W, G_S, G_Am)
    KNH_PIPE_BIG("WSA", 3, G_W, G_S, G_A)
    KNH_PIPE("WSA", 3, G_W, G_S, G_A)
    KNH_PIPE("WS", 2, G_W, G_S)
    KNH_PIPE_BIG("WmaRm", 2, G_Wma, G_Rm) // C5: modulator | carrier (in place) | mixer, 64-sample tiles
    KNH_PIPE("WmaRm", 2, G_Wma, G_Rm)     // the same with 32-sample tiles
    KNH_PIPE_FAN_("Nm", 2, G_Np, F_Nm)    // C2: phase | sin * gain on eight wavefronts | mixer
    KNH_PIPE_FAN_("N", 2, G_Np, F_N)
    KNH_PIPE("NSAm", 3, G_N, G_S, G_Am)
    KNH_PIPE("WmSDA", 3, G_Wm, G_S, G_DA)  // the delay's HBM traffic rides in the envelope wave
    KNH_PIPE("BmSA", 3, G_Bm, G_S, G_A)
    KNH_PIPE_BIG("WmEJ", 2, G_Wm, G_E2)     // many_sines: oscillator | envelope + pan (+ fold)
    KNH_PIPE("WmEJ", 2, G_Wm, G_E2)
    KNH_PIPE_BIG("WmSAJ", 3, G_Wm, G_S, G_A2)
    KNH_PIPE("WmSAJ", 3, G_Wm, G_S, G_A2)
};
const PipeEntry* KNH_PIPE_TABLE(int* n) {
  *n = (int)(sizeof(kTable) / sizeof(kTable[0]));
  return kTable;
}
#if KNH_PIPE_PART == 0
const PipeEntry* find_pipe(const char* signature, unsigned forms, int groups_per_workgroup) {
  const PipeEntry* (*const tables[])(int*) = {pipes_inplace, pipes_fold, pipes_mixer};  // preferred form first
  for (auto table : tables) {
    int n = 0;
    const PipeEntry* e = table(&n);
    for (int i = 0; i < n; ++i)
      if (std::strcmp(e[i].signature, signature) == 0 && ((forms >> e[i].form) & 1u) && e[i].gpw == groups_per_workgroup) return &e[i];
  }
  return nullptr;
}
#endif

}  // namespace knh
