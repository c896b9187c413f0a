// kernels_wide.hip -- many-wave builds of the single-wave kernel (4 or 8 voice groups per workgroup sharing one staged
// sine table) for banks with more 64-voice groups than the chip has SIMDs.  Built with -ffp-contract=off.
#include <cstring>

#include "kernel_registry.hpp"
#include "voice_pipe.hpp"

namespace knh {
using namespace knh_dev;

template <typename F, bool FMA, int WAVES, typename... S>
static hipError_t launch_wide(const VoiceKernelArgs<F>& args, unsigned n_wavefronts, hipStream_t stream) {
  if (n_wavefronts == 0) return hipSuccess;
  hipLaunchKernelGGL((voice_kernel<F, FMA, WAVES, S...>), dim3((n_wavefronts + WAVES - 1) / WAVES), dim3(WAVES * 64), 0, stream, args);
  return hipGetLastError();
}
#define KNH_WIDE(sig, ...)                                                                            \
  {sig,                                                                                               \
   {launch_wide<float, false, 4, __VA_ARGS__>, launch_wide<float, true, 4, __VA_ARGS__>},            \
   {launch_wide<float, false, 8, __VA_ARGS__>, launch_wide<float, true, 8, __VA_ARGS__>},            \
   {launch_wide<double, false, 4, __VA_ARGS__>, launch_wide<double, true, 4, __VA_ARGS__>},          \
   {launch_wide<double, false, 8, __VA_ARGS__>, launch_wide<double, true, 8, __VA_ARGS__>},          \
   {launch_wide<float, false, 16, __VA_ARGS__>, launch_wide<float, true, 16, __VA_ARGS__>},          \
   {launch_wide<double, false, 16, __VA_ARGS__>, launch_wide<double, true, 16, __VA_ARGS__>}}
static const WideEntry kWides[] = {
    KNH_WIDE("Wm", SinWt, MulVal),
    KNH_WIDE("Nm", SinNum, MulVal),
    KNH_WIDE("WmSA", SinWt, MulVal, Svf, MulAsr),
    KNH_WIDE("WmSDA", SinWt, MulVal, Svf, SampleDelay, MulAsr),
    KNH_WIDE("BmSA", PolyBlepOsc, MulVal, Svf, MulAsr),
    KNH_WIDE("WSAm", SinWt, Svf, MulAsr, MulVal),
    KNH_WIDE("WmE", SinWt, MulVal, MulAr),
    KNH_WIDE("WmaRm", SinWt, MulVal, AddVal, SinWtAr, MulVal),
    KNH_WIDE("WmEJ", SinWt, MulVal, MulAr, Pan2),
};
const WideEntry* find_wide(const char* signature) {
  for (const WideEntry& e : kWides)
    if (std::strcmp(e.signature, signature) == 0) return &e;
  return nullptr;
}

}  // namespace knh
