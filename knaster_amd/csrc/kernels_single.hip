// kernels_single.hip -- gfx950 instantiations of the single-wave fused voice-bank kernel (voice_chain.hpp), one per
// pre-built chain.  Built with -ffp-contract=off: the compiler never fuses a*b+c on its own; the FMA variants fuse explicitly.
#include <cstring>

#include "kernel_registry.hpp"
#include "voice_pipe.hpp"

namespace knh {
using namespace knh_dev;

template <typename F, bool FMA, typename... S>
static hipError_t launch_voice(const VoiceKernelArgs<F>& args, unsigned n_wavefronts, hipStream_t stream) {
  if (n_wavefronts == 0) return hipSuccess;
  hipLaunchKernelGGL((voice_kernel<F, FMA, 1, S...>), dim3(n_wavefronts), dim3(64), 0, stream, args);
  return hipGetLastError();
}

#define KNH_CHAIN(sig, ...)                                                                  \
  {sig, Chain<float, false, 0, __VA_ARGS__>::kSlots,                                         \
   {launch_voice<float, false, __VA_ARGS__>, launch_voice<float, true, __VA_ARGS__>},        \
   {launch_voice<double, false, __VA_ARGS__>, launch_voice<double, true, __VA_ARGS__>}}

static const KernelEntry kEntries[] = {
    // BASELINE.json configs
    KNH_CHAIN("Wm", SinWt, MulVal),                             // C1: SinWt * 0.2 ; bench "sine * 0.05"
    KNH_CHAIN("Nm", SinNum, MulVal),                            // C2: SinNumeric + gain
    KNH_CHAIN("WmSA", SinWt, MulVal, Svf, MulAsr),              // C3/C4: SinWt.wr_mul -> Svf -> * EnvAsr
    KNH_CHAIN("WmaRm", SinWt, MulVal, AddVal, SinWtAr, MulVal), // C5: audio-rate FM
    // single stages and common shapes
    KNH_CHAIN("W", SinWt),
    KNH_CHAIN("N", SinNum),
    KNH_CHAIN("WS", SinWt, Svf),
    KNH_CHAIN("WA", SinWt, MulAsr),
    KNH_CHAIN("WE", SinWt, MulAr),
    KNH_CHAIN("WmE", SinWt, MulVal, MulAr),                     // knaster/examples/many_sines.rs:51-63 minus Pan2
    KNH_CHAIN("WmEJ", SinWt, MulVal, MulAr, Pan2),               // knaster/examples/many_sines.rs:51-63: (env * sine.wr_mul) >> Pan2
    KNH_CHAIN("WmSAJ", SinWt, MulVal, Svf, MulAsr, Pan2),        // the C3 voice panned
    KNH_CHAIN("WSA", SinWt, Svf, MulAsr),
    KNH_CHAIN("WSAm", SinWt, Svf, MulAsr, MulVal),
    KNH_CHAIN("WLAm", SinWt, OnePoleLp, MulAsr, MulVal),
    KNH_CHAIN("WHEm", SinWt, OnePoleHp, MulAr, MulVal),
    KNH_CHAIN("NSAm", SinNum, Svf, MulAsr, MulVal),
    KNH_CHAIN("Wasd", SinWt, AddVal, SubVal, DivVal),
    KNH_CHAIN("WmV", SinWt, MulVal, MulSegEnv),                 // SinWt.wr_mul * segment Envelope
    KNH_CHAIN("WmSDA", SinWt, MulVal, Svf, SampleDelay, MulAsr), // C3 with a delay line behind the filter (HBM-bound regime)
    KNH_CHAIN("BmSA", PolyBlepOsc, MulVal, Svf, MulAsr),         // the C3 voice with a band-limited oscillator
};

const KernelEntry* find_kernel(const char* signature) {
  for (const KernelEntry& e : kEntries)
    if (std::strcmp(e.signature, signature) == 0) return &e;
  return nullptr;
}
int kernel_count() { return (int)(sizeof(kEntries) / sizeof(kEntries[0])); }
const KernelEntry* kernel_at(int i) { return (i >= 0 && i < kernel_count()) ? &kEntries[i] : nullptr; }

}  // namespace knh
