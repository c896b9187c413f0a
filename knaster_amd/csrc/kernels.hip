// kernels.hip -- gfx950 instantiations of the fused voice-bank kernel.
// Built with -ffp-contract=off: the compiler never fuses a*b+c on its own; the
// FMA variants fuse explicitly.
#include <cstring>

#include "kernel_registry.hpp"
#include "voice_dag.hpp"
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

// BIG: 64-sample tiles (32 for f64) with the fold done by the last stage group instead of a mixer wavefront
template <typename F, bool FMA, bool BIG, typename... Gs>
static hipError_t launch_pipe(const VoiceKernelArgs<F>& args, unsigned n_wavefronts, hipStream_t stream) {
  if (n_wavefronts == 0) return hipSuccess;
  constexpr int T = BIG ? PipeTile<F>::big : PipeTile<F>::value;
  hipLaunchKernelGGL((voice_pipe_kernel<F, FMA, T, BIG, Gs...>), dim3(n_wavefronts), dim3((sizeof...(Gs) + (BIG ? 0 : 1)) * 64), 0, stream, args);
  return hipGetLastError();
}
#define KNH_PIPE_AS(sig, n, big, ...)                                                               \
  {sig, n, big, {launch_pipe<float, false, big, __VA_ARGS__>, launch_pipe<float, true, big, __VA_ARGS__>}, \
   {launch_pipe<double, false, big, __VA_ARGS__>, launch_pipe<double, true, big, __VA_ARGS__>}}
#define KNH_PIPE(sig, n, ...) KNH_PIPE_AS(sig, n, false, __VA_ARGS__)
#define KNH_PIPE_BIG(sig, n, ...) KNH_PIPE_AS(sig, n, true, __VA_ARGS__)

typedef Group<SinWt, MulVal> G_Wm;
typedef Group<SinWt> G_W;
typedef Group<SinNum> G_N;
typedef Group<Svf> G_S;
typedef Group<MulAsr> G_A;
typedef Group<MulAr> G_E;
typedef Group<MulAsr, MulVal> G_Am;
typedef Group<MulVal> G_m;
typedef Group<SinWt, MulVal, AddVal> G_Wma;
typedef Group<SinWtAr, MulVal> G_Rm;
typedef Group<SampleDelay, MulAsr> G_DA;
typedef Group<PolyBlepOsc, MulVal> G_Bm;

static const PipeEntry kPipes[] = {
    KNH_PIPE_BIG("WmSA", 3, G_Wm, G_S, G_A),   // C3/C4: oscillator | filter | envelope + fold, 64-sample tiles
    KNH_PIPE("WmSA", 3, G_Wm, G_S, G_A),       // the same with 32-sample tiles and a mixer wavefront (KNH_PIPE_BIG=0)
    KNH_PIPE_BIG("WSAm", 3, G_W, G_S, G_Am),
    KNH_PIPE("WSAm", 3, G_W, G_S, G_Am),
    KNH_PIPE_BIG("WSA", 3, G_W, G_S, G_A),
    KNH_PIPE("WSA", 3, G_W, G_S, G_A),
    KNH_PIPE("WS", 2, G_W, G_S),
    KNH_PIPE("WmaRm", 2, G_Wma, G_Rm),     // C5: modulator | carrier
    KNH_PIPE("Nm", 2, G_N, G_m),           // C2
    KNH_PIPE("NSAm", 3, G_N, G_S, G_Am),
    KNH_PIPE("WmSDA", 3, G_Wm, G_S, G_DA),  // the delay's HBM traffic rides in the envelope wave
    KNH_PIPE("BmSA", 3, G_Bm, G_S, G_A),
};
const PipeEntry* find_pipe(const char* signature, bool allow_big) {
  for (const PipeEntry& e : kPipes)
    if (std::strcmp(e.signature, signature) == 0 && (allow_big || !e.big)) return &e;
  return nullptr;
}

template <bool FMA, bool AR, typename SRC, typename POST>
static hipError_t launch_dag(const VoiceKernelArgs<float>& args, unsigned n_wavefronts, hipStream_t stream) {
  if (n_wavefronts == 0) return hipSuccess;
  hipLaunchKernelGGL((voice_dag_kernel<float, FMA, AR, SRC, POST>), dim3(n_wavefronts), dim3(320), 0, stream, args);
  return hipGetLastError();
}
#define KNH_DAG(sig, ar, src, post) {sig, {launch_dag<false, ar, src, post>, launch_dag<true, ar, src, post>}}
typedef Group<> G_none;
static const DagEntry kDags[] = {
    KNH_DAG("WmSA", false, G_Wm, G_none),  // C3
    KNH_DAG("WSA", false, G_W, G_none),
    KNH_DAG("WSAm", false, G_W, G_m),
    KNH_DAG("NSAm", false, G_N, G_m),
};
const DagEntry* find_dag(const char* signature) {
  for (const DagEntry& e : kDags)
    if (std::strcmp(e.signature, signature) == 0) return &e;
  return nullptr;
}

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
   {launch_wide<double, false, 8, __VA_ARGS__>, launch_wide<double, true, 8, __VA_ARGS__>}}
static const WideEntry kWides[] = {
    KNH_WIDE("Wm", SinWt, MulVal),
    KNH_WIDE("Nm", SinNum, MulVal),
    KNH_WIDE("WmSA", SinWt, MulVal, Svf, MulAsr),
    KNH_WIDE("WmSDA", SinWt, MulVal, Svf, SampleDelay, MulAsr),
    KNH_WIDE("BmSA", PolyBlepOsc, MulVal, Svf, MulAsr),
    KNH_WIDE("WSAm", SinWt, Svf, MulAsr, MulVal),
    KNH_WIDE("WmE", SinWt, MulVal, MulAr),
    KNH_WIDE("WmaRm", SinWt, MulVal, AddVal, SinWtAr, MulVal),
};
const WideEntry* find_wide(const char* signature) {
  for (const WideEntry& e : kWides)
    if (std::strcmp(e.signature, signature) == 0) return &e;
  return nullptr;
}

const KernelEntry* find_kernel(const char* signature) {
  for (const KernelEntry& e : kEntries)
    if (std::strcmp(e.signature, signature) == 0) return &e;
  return nullptr;
}
int kernel_count() { return (int)(sizeof(kEntries) / sizeof(kEntries[0])); }
const KernelEntry* kernel_at(int i) { return (i >= 0 && i < kernel_count()) ? &kEntries[i] : nullptr; }

template <typename F>
static hipError_t launch_fold(bool tree, const F* rows, unsigned n_rows, unsigned row_len, unsigned frame_begin, unsigned frame_end,
                              F* out, unsigned channels, unsigned out_stride, unsigned n_blocks, bool accumulate, unsigned* zero_flags, hipStream_t s) {
  if (frame_end <= frame_begin || n_rows == 0 || n_blocks == 0)  // nothing to fold: the flag words still have to be cleared
    return zero_flags ? hipMemsetAsync(zero_flags, 0, 2 * sizeof(unsigned), s) : hipSuccess;
  if (tree) {
    unsigned grid = (frame_end - frame_begin + 15u) / 16u;
    hipLaunchKernelGGL((fold_tree_kernel<F>), dim3(grid, n_blocks), dim3(256), 0, s, rows, n_rows, row_len, frame_begin, frame_end,
                       out, channels, out_stride, accumulate ? 1u : 0u, zero_flags);
  } else {
    unsigned grid = (frame_end - frame_begin + 63u) / 64u;
    hipLaunchKernelGGL((fold_rows_kernel<F>), dim3(grid, n_blocks), dim3(64), 0, s, rows, n_rows, row_len, frame_begin, frame_end,
                       out, channels, out_stride, accumulate ? 1u : 0u, zero_flags);
  }
  return hipGetLastError();
}
hipError_t launch_fold_f32(bool tree, const float* rows, unsigned n_rows, unsigned row_len, unsigned frame_begin,
                           unsigned frame_end, float* out, unsigned channels, unsigned out_stride, unsigned n_blocks, bool accumulate,
                           unsigned* zero_flags, hipStream_t s) {
  return launch_fold<float>(tree, rows, n_rows, row_len, frame_begin, frame_end, out, channels, out_stride, n_blocks, accumulate, zero_flags, s);
}
hipError_t launch_fold_f64(bool tree, const double* rows, unsigned n_rows, unsigned row_len, unsigned frame_begin,
                           unsigned frame_end, double* out, unsigned channels, unsigned out_stride, unsigned n_blocks, bool accumulate,
                           unsigned* zero_flags, hipStream_t s) {
  return launch_fold<double>(tree, rows, n_rows, row_len, frame_begin, frame_end, out, channels, out_stride, n_blocks, accumulate, zero_flags, s);
}


template <typename F>
__global__ void __launch_bounds__(256) sum_shards_kernel(const F* shards, unsigned n_shards, size_t shard_stride, size_t n,
                                                         unsigned block_size, unsigned frame_begin, unsigned frame_end, F* out,
                                                         unsigned accumulate) {
  const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const unsigned frame = (unsigned)(i % block_size);
  if (frame < frame_begin || frame >= frame_end) return;
  F acc = shards[i];
  for (unsigned k = 1; k < n_shards; ++k) acc = acc + shards[k * shard_stride + i];
  out[i] = accumulate ? out[i] + acc : acc;
}
template <typename F>
static hipError_t launch_sum_shards(const F* shards, unsigned n_shards, size_t shard_stride, size_t n, unsigned block_size,
                                    unsigned frame_begin, unsigned frame_end, F* out, bool accumulate, hipStream_t s) {
  if (n == 0 || n_shards == 0 || frame_end <= frame_begin) return hipSuccess;
  hipLaunchKernelGGL((sum_shards_kernel<F>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, shards, n_shards, shard_stride, n,
                     block_size, frame_begin, frame_end, out, accumulate ? 1u : 0u);
  return hipGetLastError();
}
hipError_t launch_sum_shards_f32(const float* shards, unsigned n_shards, size_t shard_stride, size_t n, unsigned block_size,
                                 unsigned frame_begin, unsigned frame_end, float* out, bool accumulate, hipStream_t s) {
  return launch_sum_shards<float>(shards, n_shards, shard_stride, n, block_size, frame_begin, frame_end, out, accumulate, s);
}
hipError_t launch_sum_shards_f64(const double* shards, unsigned n_shards, size_t shard_stride, size_t n, unsigned block_size,
                                 unsigned frame_begin, unsigned frame_end, double* out, bool accumulate, hipStream_t s) {
  return launch_sum_shards<double>(shards, n_shards, shard_stride, n, block_size, frame_begin, frame_end, out, accumulate, s);
}

}  // namespace knh
