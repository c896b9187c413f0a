// kernels_fold.hip -- the mix kernels (tree fold, exact left fold, sum of host shards) and, only when the library is
// built with KNH_BUILD_DAG=1, the experimental five-role pipeline of voice_dag.hpp (measured slower than the linear
// pipeline; not part of the default build).
#include <cstring>

#include "kernel_registry.hpp"
#ifdef KNH_WITH_DAG
#include "voice_dag.hpp"
#endif
#include "voice_pipe.hpp"

namespace knh {
using namespace knh_dev;

#ifdef KNH_WITH_DAG
template <bool FMA, bool AR, typename SRC, typename POST>
static hipError_t launch_dag(const VoiceKernelArgs<float>& args, unsigned n_wavefronts, hipStream_t stream) {
  if (n_wavefronts == 0) return hipSuccess;
  hipLaunchKernelGGL((voice_dag_kernel<float, FMA, AR, SRC, POST>), dim3(n_wavefronts), dim3(320), 0, stream, args);
  return hipGetLastError();
}
#define KNH_DAG(sig, ar, src, post) {sig, {launch_dag<false, ar, src, post>, launch_dag<true, ar, src, post>}}
typedef Group<> G_none;
typedef Group<SinWt, MulVal> G_Wm;
typedef Group<SinWt> G_W;
typedef Group<SinNum> G_N;
typedef Group<MulVal> G_m;
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
#else
const DagEntry* find_dag(const char*) { return nullptr; }
#endif

template <typename F>
static hipError_t launch_fold(bool tree, const F* rows, unsigned n_rows, unsigned row_len, unsigned frame_begin, unsigned frame_end,
                              F* out, unsigned channels, unsigned out_stride, unsigned n_blocks, bool accumulate, unsigned* zero_flags,
                              const HostDone& host, hipStream_t s) {
  if (frame_end <= frame_begin || n_rows == 0 || n_blocks == 0) {  // nothing to fold: the flag words still have to be cleared
    if (host.words) return hipErrorInvalidValue;  // (the caller hands a launch over this way only when there is something to fold)
    return zero_flags ? hipMemsetAsync(zero_flags, 0, 2 * sizeof(unsigned), s) : hipSuccess;
  }
  if (tree) {
    unsigned grid = (frame_end - frame_begin + 15u) / 16u;
    hipLaunchKernelGGL((fold_tree_kernel<F>), dim3(grid, n_blocks), dim3(256), 0, s, rows, n_rows, row_len, frame_begin, frame_end,
                       out, channels, out_stride, accumulate ? 1u : 0u, zero_flags, host);
  } else {
    unsigned grid = (frame_end - frame_begin + 63u) / 64u;
    hipLaunchKernelGGL((fold_rows_kernel<F>), dim3(grid, n_blocks), dim3(64), 0, s, rows, n_rows, row_len, frame_begin, frame_end,
                       out, channels, out_stride, accumulate ? 1u : 0u, zero_flags, host);
  }
  return hipGetLastError();
}
hipError_t launch_fold_f32(bool tree, const float* rows, unsigned n_rows, unsigned row_len, unsigned frame_begin,
                           unsigned frame_end, float* out, unsigned channels, unsigned out_stride, unsigned n_blocks, bool accumulate,
                           unsigned* zero_flags, hipStream_t s, const knh_dev::HostDone* host) {
  return launch_fold<float>(tree, rows, n_rows, row_len, frame_begin, frame_end, out, channels, out_stride, n_blocks, accumulate, zero_flags,
                            host ? *host : HostDone{nullptr, nullptr, nullptr, 0u}, s);
}
hipError_t launch_fold_f64(bool tree, const double* rows, unsigned n_rows, unsigned row_len, unsigned frame_begin,
                           unsigned frame_end, double* out, unsigned channels, unsigned out_stride, unsigned n_blocks, bool accumulate,
                           unsigned* zero_flags, hipStream_t s, const knh_dev::HostDone* host) {
  return launch_fold<double>(tree, rows, n_rows, row_len, frame_begin, frame_end, out, channels, out_stride, n_blocks, accumulate, zero_flags,
                             host ? *host : HostDone{nullptr, nullptr, nullptr, 0u}, s);
}


// the fold server of a resident launch (voice_chain.hpp, res_fold_server): four wavefronts per 32 partial rows + four for the root
hipError_t launch_res_server_f32(const knh_dev::ResServerArgs<float>& a, hipStream_t s) {
  hipLaunchKernelGGL((res_fold_server<float>), dim3((a.n_rows + 31u) / 32u + 1u), dim3(256), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_res_server_f64(const knh_dev::ResServerArgs<double>& a, hipStream_t s) {
  hipLaunchKernelGGL((res_fold_server<double>), dim3((a.n_rows + 31u) / 32u + 1u), dim3(256), 0, s, a);
  return hipGetLastError();
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
