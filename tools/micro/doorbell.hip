// What does it cost to hand a command to a kernel that is already RESIDENT on every CU, and to get its answer back?
// (The per-block call of the boundary -- UGen::process_block once per block, knaster_graph/src/task.rs:25-31 -- pays a kernel
// launch, the sine table's staging and the pipeline's fill today; a resident kernel would pay this instead.)
//
// One workgroup of 256 threads per CU (150 KiB of LDS each, so that no two share a CU), all of them waiting for an epoch word:
//   A  every workgroup polls the word in mapped pinned HOST memory (256 readers over PCIe)
//   B  workgroup 0 polls the host word and republishes it in a device word the others poll (one PCIe reader + one hop)
//   C  the word lives in fine-grained DEVICE memory that the host writes through the BAR (if the host can address it)
// Each round: the host stores epoch k; every workgroup that has seen it arrives on a counter (one counter of 256 arrivals, or
// 32 + 8 in two levels: 32 neighbours, then the 8 group leaders); the last arriver stores k into a host word the host polls.
// Every spin is bounded by the constant-rate clock (s_memrealtime, 100 MHz): a kernel whose host went away exits by itself.
//   L  for comparison: one LAUNCH per round of a 256-workgroup kernel with the same fan-in and the same host word.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(2); } } while (0)

struct Ctl {
  const uint32_t* bell;   // the epoch word the host rings (host or device memory)
  uint32_t* relay;        // device word (mode B)
  uint32_t* counters;     // device: [0] top, [16 * (1 + g)] group g (one 64-byte line each)
  uint32_t* done;         // mapped pinned host memory: epoch of the last finished round
  uint32_t rounds;
  uint32_t mode;          // 0 A, 1 B, 2 C
  uint32_t two_level;
  uint64_t timeout_ticks;  // s_memrealtime ticks (10 ns)
};

__device__ __forceinline__ uint32_t load_sys(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ uint32_t load_dev(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// arrive; true for the last arriver of the whole grid
__device__ __forceinline__ bool arrive(const Ctl& c, uint32_t wg, uint32_t n_wg) {
  if (!c.two_level) return atomicAdd(&c.counters[0], 1u) % n_wg == n_wg - 1u;
  const uint32_t g = wg / 32u, in_group = (n_wg - g * 32u) < 32u ? n_wg - g * 32u : 32u, groups = (n_wg + 31u) / 32u;
  if (atomicAdd(&c.counters[16u * (1u + g)], 1u) % in_group != in_group - 1u) return false;
  return atomicAdd(&c.counters[0], 1u) % groups == groups - 1u;
}

__global__ void __launch_bounds__(256) resident(Ctl c) {
  extern __shared__ uint32_t lds[];
  const uint32_t wg = blockIdx.x, n_wg = gridDim.x;
  for (uint32_t k = 1; k <= c.rounds; ++k) {
    if (threadIdx.x == 0) {
      const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
      uint32_t seen = 0;
      for (;;) {
        if (c.mode == 1u && wg != 0u) seen = load_dev(c.relay);
        else if (c.mode == 2u) seen = load_dev(c.bell);
        else seen = load_sys(c.bell);
        if (seen >= k || __builtin_amdgcn_s_memrealtime() - t0 > c.timeout_ticks) break;
        __builtin_amdgcn_s_sleep(2);
      }
      if (c.mode == 1u && wg == 0u && seen >= k) __hip_atomic_store(c.relay, seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      lds[0] = seen >= k ? 1u : 0u;
    }
    __syncthreads();
    const bool go = lds[0] != 0u;
    __syncthreads();
    if (!go) return;  // the host went away: every workgroup times out on its own and leaves
    if (threadIdx.x == 0 && arrive(c, wg, n_wg)) __hip_atomic_store(c.done, k, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

__global__ void __launch_bounds__(256) one_round(Ctl c, uint32_t k) {
  if (threadIdx.x == 0 && arrive(c, blockIdx.x, gridDim.x)) __hip_atomic_store(c.done, k, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

static bool wait_done(volatile uint32_t* done, uint32_t k, double limit_s) {
  const auto t0 = std::chrono::steady_clock::now();
  for (uint64_t spin = 0;; ++spin) {
    if (__atomic_load_n(done, __ATOMIC_ACQUIRE) >= k) return true;
    if ((spin & 0xFFFF) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit_s) return false;
    __builtin_ia32_pause();
  }
}

static void report(const char* name, std::vector<double>& us) {
  std::sort(us.begin(), us.end());
  double sum = 0;
  for (double x : us) sum += x;
  std::printf("%-58s  min %6.2f  p50 %6.2f  mean %6.2f  p99 %6.2f us   (%zu rounds)\n", name, us.front(), us[us.size() / 2], sum / us.size(), us[size_t(us.size() * 0.99)], us.size());
  std::fflush(stdout);
}

int main(int argc, char** argv) {
  const uint32_t rounds = 4000;
  int n_cu = 0;
  CK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0));
  const uint32_t n_wg = argc > 1 ? std::atoi(argv[1]) : (uint32_t)n_cu;
  std::printf("%d CUs, %u workgroups of 256 threads, 150 KiB of LDS each\n", n_cu, n_wg);
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(resident), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  uint32_t *h_bell = nullptr, *h_done = nullptr, *d_relay = nullptr, *d_counters = nullptr, *d_bell = nullptr;
  CK(hipHostMalloc(&h_bell, 64, hipHostMallocMapped | hipHostMallocCoherent));
  CK(hipHostMalloc(&h_done, 64, hipHostMallocMapped | hipHostMallocCoherent));
  CK(hipMalloc(&d_relay, 64));
  CK(hipMalloc(&d_counters, 64 * 16));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  // C: fine-grained device memory the host may be able to write through the BAR.  Probed BEFORE any kernel is in flight: if
  // the store faults, nothing is left running on the device.
  bool have_c = false;
  if (hipExtMallocWithFlags(reinterpret_cast<void**>(&d_bell), 64, hipDeviceMallocFinegrained) == hipSuccess) {
    hipPointerAttribute_t at{};
    const bool attr = hipPointerGetAttributes(&at, d_bell) == hipSuccess;
    std::printf("fine-grained device word %p: attributes %s, hostPointer %p, devicePointer %p\n", (void*)d_bell, attr ? "ok" : "none", attr ? at.hostPointer : nullptr, attr ? at.devicePointer : nullptr);
    if (std::getenv("DOORBELL_TRY_BAR")) {
      CK(hipMemset(d_bell, 0, 64));
      CK(hipDeviceSynchronize());
      std::printf("storing to it from the host ...\n");
      std::fflush(stdout);
      *reinterpret_cast<volatile uint32_t*>(d_bell) = 0x1234u;  // (a fault here ends the process with the device idle)
      __builtin_ia32_sfence();
      uint32_t back = 0;
      CK(hipMemcpy(&back, d_bell, 4, hipMemcpyDeviceToHost));
      std::printf("read back %#x\n", back);
      have_c = back == 0x1234u;
    }
  } else {
    std::printf("hipExtMallocWithFlags(hipDeviceMallocFinegrained) failed\n");
  }

  for (uint32_t two = 0; two < 2; ++two) {
    for (uint32_t mode = 0; mode < 3; ++mode) {
      if (mode == 2 && !have_c) continue;
      uint32_t* bell = mode == 2 ? d_bell : h_bell;
      *h_done = 0;
      if (mode == 2) { CK(hipMemset(d_bell, 0, 64)); } else { *h_bell = 0; }
      CK(hipMemset(d_relay, 0, 64));
      CK(hipMemset(d_counters, 0, 64 * 16));
      CK(hipDeviceSynchronize());
      Ctl c{bell, d_relay, d_counters, h_done, rounds, mode, two, 20u * 100000u /* 20 ms */};
      hipLaunchKernelGGL(resident, dim3(n_wg), dim3(256), 150 * 1024, s, c);
      CK(hipGetLastError());
      std::vector<double> us;
      bool ok = true;
      for (uint32_t k = 1; k <= rounds && ok; ++k) {
        // a little host work between rounds, as a caller would have (and so that the pollers are in their steady spin)
        const auto tw = std::chrono::steady_clock::now();
        while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tw).count() < 3.0) {}
        const auto t0 = std::chrono::steady_clock::now();
        __atomic_store_n(reinterpret_cast<volatile uint32_t*>(bell), k, __ATOMIC_RELEASE);
        if (mode == 2) __builtin_ia32_sfence();
        ok = wait_done(h_done, k, 0.5);
        us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
      }
      CK(hipStreamSynchronize(s));  // (bounded: every workgroup leaves after 20 ms without a command)
      if (!ok) { std::printf("mode %u two_level %u: a round was not answered within 0.5 s\n", mode, two); continue; }
      char name[128];
      std::snprintf(name, sizeof name, "resident, %s, fan-in %s", mode == 0 ? "A all poll host word" : mode == 1 ? "B leader polls host, relays" : "C host writes device word (BAR)",
                    two ? "32 + 8" : "256 on one counter");
      us.erase(us.begin(), us.begin() + 200);
      report(name, us);
    }
  }
  for (uint32_t two = 0; two < 2; ++two) {
    *h_done = 0;
    CK(hipMemset(d_counters, 0, 64 * 16));
    CK(hipDeviceSynchronize());
    Ctl c{h_bell, d_relay, d_counters, h_done, rounds, 0, two, 0};
    std::vector<double> us;
    for (uint32_t k = 1; k <= rounds; ++k) {
      const auto tw = std::chrono::steady_clock::now();
      while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tw).count() < 3.0) {}
      const auto t0 = std::chrono::steady_clock::now();
      hipLaunchKernelGGL(one_round, dim3(n_wg), dim3(256), 0, s, c, k);
      if (!wait_done(h_done, k, 0.5)) { std::printf("launch round not answered\n"); return 1; }
      us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
    CK(hipStreamSynchronize(s));
    us.erase(us.begin(), us.begin() + 200);
    report(two ? "one LAUNCH per round (no LDS), fan-in 32 + 8" : "one LAUNCH per round (no LDS), fan-in 256 on one counter", us);
  }
  return 0;
}
