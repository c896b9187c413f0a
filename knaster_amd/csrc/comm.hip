// comm.hip -- the one collective of the path: the sum of the per-GPU stereo blocks, on RCCL over xGMI.
//
// One process per GPU.  The voices of a graph are independent, so the only exchange between ranks is
// `ncclReduce(sum, root)` of a launch's mixed blocks ([n_blocks][channels][block_size] samples, 256 KiB for 64 stereo
// f32 blocks): the additive graph output of knaster_graph/src/graph.rs:827-872 continued across GPUs.  RCCL is loaded at
// run time (dlopen of librccl.so.1: the copy a host process already holds -- PyTorch ships one -- or ROCm's), so that a
// single-GPU host needs no RCCL at all; a process that asks for a communicator without one gets KNH_ERR_DEVICE.
//
// The reduce runs on the communicator's own stream, ordered after the producer stream by an event, so the next launch's
// kernels overlap it; knh_comm_wait / knh_comm_synchronize order a consumer after it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "../../include/knaster_hip.h"

namespace {

struct Rccl {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclReduce) Reduce = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;
  std::string error;
};

Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (r.lib) break;
    }
    if (!r.lib) { r.error = std::string("RCCL not found (dlopen librccl.so.1): ") + (dlerror() ? dlerror() : ""); return; }
    auto sym = [&](const char* name) {
      void* p = dlsym(r.lib, name);
      if (!p && r.error.empty()) r.error = std::string("RCCL symbol missing: ") + name;
      return p;
    };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommCount = reinterpret_cast<decltype(r.CommCount)>(sym("ncclCommCount"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.Reduce = reinterpret_cast<decltype(r.Reduce)>(sym("ncclReduce"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(sym("ncclGetVersion"));
  });
  return &r;
}

std::string g_comm_create_error;

}  // namespace

struct knh_comm {
  ncclComm_t comm = nullptr;
  int device = 0;
  uint32_t rank = 0, world = 1;
  hipStream_t stream = nullptr;   // the reduces run here
  hipEvent_t produced = nullptr;  // recorded on the producer stream, waited on by `stream`
  hipEvent_t reduced = nullptr;   // recorded on `stream` after the last reduce
  bool pending = false;
  // the last reduce of each of the buffers seen lately (a host alternates two, so that a launch overlaps the reduce of
  // the one before): the buffer's next producer waits for exactly that one
  struct Slot { void* buf = nullptr; hipEvent_t done = nullptr; uint64_t stamp = 0; };
  Slot slots[4];
  uint64_t clock = 0;
  // measurement (bench.py): device time of the reduces since the last reset, from HIP events on the communicator's stream
  bool timing = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> timing_pool;
  size_t timing_used = 0;
  double timing_ms = 0.0;
  uint64_t timing_count = 0;
  std::string err;
  int fail(int code, const std::string& m) { err = m; return code; }
};

// no C++ exception crosses the C ABI (bank.hip has the same guard for the bank entry points)
namespace {
int comm_timing_collect(knh_comm* c) {
  if (hipStreamSynchronize(c->stream) != hipSuccess) return c->fail(KNH_ERR_DEVICE, "hipStreamSynchronize failed");
  for (size_t k = 0; k < c->timing_used; ++k) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, c->timing_pool[k].first, c->timing_pool[k].second) != hipSuccess) return c->fail(KNH_ERR_DEVICE, "hipEventElapsedTime failed");
    c->timing_ms += ms;
    c->timing_count += 1;
  }
  c->timing_used = 0;
  return KNH_OK;
}
template <typename Fn>
int32_t comm_guarded(knh_comm* c, Fn&& fn) noexcept {
  try {
    return fn();
  } catch (const std::bad_alloc&) {
    try { (c ? c->err : g_comm_create_error) = "out of host memory (std::bad_alloc)"; } catch (...) {}
    return KNH_ERR_OUT_OF_MEMORY;
  } catch (...) {
    try { (c ? c->err : g_comm_create_error) = "internal error"; } catch (...) {}
    return KNH_ERR_INTERNAL;
  }
}
}  // namespace

static_assert(sizeof(ncclUniqueId) == KNH_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");

extern "C" {

int32_t knh_comm_unique_id(uint8_t* id) {
  return comm_guarded(nullptr, [&]() -> int32_t {
    if (!id) { g_comm_create_error = "null id"; return KNH_ERR_INVALID_ARGUMENT; }
    Rccl* r = rccl();
    if (!r->error.empty()) { g_comm_create_error = r->error; return KNH_ERR_DEVICE; }
    ncclUniqueId u;
    ncclResult_t rc = r->GetUniqueId(&u);
    if (rc != ncclSuccess) { g_comm_create_error = std::string("ncclGetUniqueId: ") + r->GetErrorString(rc); return KNH_ERR_DEVICE; }
    std::memcpy(id, &u, KNH_COMM_ID_BYTES);
    return KNH_OK;
  });
}

int32_t knh_comm_create(uint32_t rank, uint32_t world, const uint8_t* id, int32_t device, knh_comm** out) {
  return comm_guarded(nullptr, [&]() -> int32_t {
    if (out) *out = nullptr;
    if (!out || !id || world == 0 || rank >= world) { g_comm_create_error = "bad rank/world/id"; return KNH_ERR_INVALID_ARGUMENT; }
    Rccl* r = rccl();
    if (!r->error.empty()) { g_comm_create_error = r->error; return KNH_ERR_DEVICE; }
    auto c = new knh_comm();
    auto bail = [&](const std::string& m) { g_comm_create_error = m; knh_comm_destroy(c); return KNH_ERR_DEVICE; };
    if (device >= 0) c->device = device;
    else if (hipGetDevice(&c->device) != hipSuccess) return bail("hipGetDevice failed");
    if (hipSetDevice(c->device) != hipSuccess) return bail("hipSetDevice failed");
    c->rank = rank;
    c->world = world;
    ncclUniqueId u;
    std::memcpy(&u, id, KNH_COMM_ID_BYTES);
    ncclResult_t rc = r->CommInitRank(&c->comm, static_cast<int>(world), u, static_cast<int>(rank));
    if (rc != ncclSuccess) { c->comm = nullptr; return bail(std::string("ncclCommInitRank: ") + r->GetErrorString(rc)); }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return bail("hipStreamCreate failed");
    if (hipEventCreateWithFlags(&c->produced, hipEventDisableTiming) != hipSuccess) return bail("hipEventCreate failed");
    if (hipEventCreateWithFlags(&c->reduced, hipEventDisableTiming) != hipSuccess) return bail("hipEventCreate failed");
    *out = c;
    return KNH_OK;
  });
}

void knh_comm_destroy(knh_comm* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm) (void)rccl()->CommDestroy(c->comm);
  if (c->produced) (void)hipEventDestroy(c->produced);
  if (c->reduced) (void)hipEventDestroy(c->reduced);
  for (auto& sl : c->slots)
    if (sl.done) (void)hipEventDestroy(sl.done);
  for (auto& p : c->timing_pool) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

const char* knh_comm_last_error(const knh_comm* c) { return c ? c->err.c_str() : g_comm_create_error.c_str(); }

uint32_t knh_comm_world(const knh_comm* c) {
  if (!c || !c->comm) return 0;
  int n = 0;
  if (rccl()->CommCount(c->comm, &n) != ncclSuccess) return 0;
  return static_cast<uint32_t>(n);
}

int32_t knh_comm_rccl_version(void) {
  Rccl* r = rccl();
  int v = 0;
  if (!r->error.empty() || !r->GetVersion || r->GetVersion(&v) != ncclSuccess) return 0;
  return v;
}

int32_t knh_comm_reduce_sum(knh_comm* c, void* buf, size_t count, uint32_t sample_type, uint32_t root, void* after_stream) {
  return comm_guarded(c, [&]() -> int32_t {
    if (!c) return KNH_ERR_INVALID_ARGUMENT;
    if (!buf || root >= c->world || sample_type > KNH_F64) return c->fail(KNH_ERR_INVALID_ARGUMENT, "knh_comm_reduce_sum: bad argument");
    if (hipSetDevice(c->device) != hipSuccess) return c->fail(KNH_ERR_DEVICE, "hipSetDevice failed");
    // everything the producer stream has been given so far comes first
    hipStream_t producer = static_cast<hipStream_t>(after_stream);
    if (hipEventRecord(c->produced, producer) != hipSuccess) return c->fail(KNH_ERR_DEVICE, "hipEventRecord failed");
    if (hipStreamWaitEvent(c->stream, c->produced, 0) != hipSuccess) return c->fail(KNH_ERR_DEVICE, "hipStreamWaitEvent failed");
    std::pair<hipEvent_t, hipEvent_t>* tp = nullptr;
    if (c->timing) {
      if (c->timing_used == c->timing_pool.size()) {
        if (c->timing_pool.size() >= 4096) {
          int r2 = comm_timing_collect(c);
          if (r2 != KNH_OK) return r2;
        } else {
          hipEvent_t e0, e1;
          if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return c->fail(KNH_ERR_DEVICE, "hipEventCreate failed");
          c->timing_pool.emplace_back(e0, e1);
        }
      }
      tp = &c->timing_pool[c->timing_used++];
      if (hipEventRecord(tp->first, c->stream) != hipSuccess) return c->fail(KNH_ERR_DEVICE, "hipEventRecord failed");
    }
    ncclResult_t rc = rccl()->Reduce(buf, buf, count, sample_type == KNH_F64 ? ncclFloat64 : ncclFloat32, ncclSum, static_cast<int>(root), c->comm, c->stream);
    if (rc != ncclSuccess) return c->fail(KNH_ERR_DEVICE, std::string("ncclReduce: ") + rccl()->GetErrorString(rc));
    if (tp && hipEventRecord(tp->second, c->stream) != hipSuccess) return c->fail(KNH_ERR_DEVICE, "hipEventRecord failed");
    if (hipEventRecord(c->reduced, c->stream) != hipSuccess) return c->fail(KNH_ERR_DEVICE, "hipEventRecord failed");
    c->pending = true;
    knh_comm::Slot* slot = nullptr;
    for (auto& sl : c->slots)
      if (sl.buf == buf) slot = &sl;
    if (!slot) {  // the least recently used one; an event recorded later on the same stream covers what it stood for
      slot = &c->slots[0];
      for (auto& sl : c->slots)
        if (sl.stamp < slot->stamp) slot = &sl;
    }
    if (!slot->done && hipEventCreateWithFlags(&slot->done, hipEventDisableTiming) != hipSuccess) return c->fail(KNH_ERR_DEVICE, "hipEventCreate failed");
    slot->buf = buf;
    slot->stamp = ++c->clock;
    if (hipEventRecord(slot->done, c->stream) != hipSuccess) return c->fail(KNH_ERR_DEVICE, "hipEventRecord failed");
    return KNH_OK;
  });
}

// The minimum over all ranks of one status word (1 = this rank is fine), on the communicator's stream, waited for: how the
// ranks of a bank agree at the end of knh_bank_init that EVERY rank's own voices came up -- a rank that failed and left
// would otherwise leave the others blocked in their first ncclReduce.  Not part of the C ABI (rank_bank.hpp calls it).
int32_t knh_comm_all_min(knh_comm* c, int32_t value, int32_t* out) {
  return comm_guarded(c, [&]() -> int32_t {
    if (!c || !out) return KNH_ERR_INVALID_ARGUMENT;
    *out = value;
    if (c->world <= 1) return KNH_OK;
    if (hipSetDevice(c->device) != hipSuccess) return c->fail(KNH_ERR_DEVICE, "hipSetDevice failed");
    int32_t* d = nullptr;
    if (hipMalloc(&d, sizeof(int32_t)) != hipSuccess) return c->fail(KNH_ERR_DEVICE, "hipMalloc failed");
    int32_t rc_out = KNH_OK;
    if (hipMemcpyAsync(d, &value, sizeof value, hipMemcpyHostToDevice, c->stream) != hipSuccess) rc_out = c->fail(KNH_ERR_DEVICE, "hipMemcpyAsync failed");
    if (rc_out == KNH_OK) {
      const ncclResult_t rc = rccl()->AllReduce(d, d, 1, ncclInt32, ncclMin, c->comm, c->stream);
      if (rc != ncclSuccess) rc_out = c->fail(KNH_ERR_DEVICE, std::string("ncclAllReduce: ") + rccl()->GetErrorString(rc));
    }
    if (rc_out == KNH_OK && (hipMemcpyAsync(out, d, sizeof value, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess))
      rc_out = c->fail(KNH_ERR_DEVICE, "reading the agreed status failed");
    (void)hipFree(d);
    return rc_out;
  });
}
int32_t knh_comm_wait_buffer(knh_comm* c, const void* buf, void* stream) {
  return comm_guarded(c, [&]() -> int32_t {
    if (!c) return KNH_ERR_INVALID_ARGUMENT;
    if (!c->pending) return KNH_OK;
    if (hipSetDevice(c->device) != hipSuccess) return c->fail(KNH_ERR_DEVICE, "hipSetDevice failed");
    for (auto& sl : c->slots)
      if (sl.buf == buf && sl.done)
        return hipStreamWaitEvent(static_cast<hipStream_t>(stream), sl.done, 0) == hipSuccess ? KNH_OK : c->fail(KNH_ERR_DEVICE, "hipStreamWaitEvent failed");
    // not among the recent ones: every reduce so far
    return hipStreamWaitEvent(static_cast<hipStream_t>(stream), c->reduced, 0) == hipSuccess ? KNH_OK : c->fail(KNH_ERR_DEVICE, "hipStreamWaitEvent failed");
  });
}

int32_t knh_comm_wait(knh_comm* c, void* stream) {
  return comm_guarded(c, [&]() -> int32_t {
    if (!c) return KNH_ERR_INVALID_ARGUMENT;
    if (!c->pending) return KNH_OK;
    if (hipSetDevice(c->device) != hipSuccess) return c->fail(KNH_ERR_DEVICE, "hipSetDevice failed");
    if (hipStreamWaitEvent(static_cast<hipStream_t>(stream), c->reduced, 0) != hipSuccess) return c->fail(KNH_ERR_DEVICE, "hipStreamWaitEvent failed");
    return KNH_OK;
  });
}

int32_t knh_comm_timing_reset(knh_comm* c, int32_t enable) {
  return comm_guarded(c, [&]() -> int32_t {
    if (!c) return KNH_ERR_INVALID_ARGUMENT;
    if (hipSetDevice(c->device) != hipSuccess) return c->fail(KNH_ERR_DEVICE, "hipSetDevice failed");
    int rc = comm_timing_collect(c);
    if (rc != KNH_OK) return rc;
    c->timing_ms = 0.0;
    c->timing_count = 0;
    c->timing = enable != 0;
    return KNH_OK;
  });
}

int32_t knh_comm_timing_read(knh_comm* c, double* reduce_ms, uint64_t* reduces) {
  return comm_guarded(c, [&]() -> int32_t {
    if (!c) return KNH_ERR_INVALID_ARGUMENT;
    if (hipSetDevice(c->device) != hipSuccess) return c->fail(KNH_ERR_DEVICE, "hipSetDevice failed");
    int rc = comm_timing_collect(c);
    if (rc != KNH_OK) return rc;
    if (reduce_ms) *reduce_ms = c->timing_ms;
    if (reduces) *reduces = c->timing_count;
    return KNH_OK;
  });
}

int32_t knh_comm_synchronize(knh_comm* c) {
  return comm_guarded(c, [&]() -> int32_t {
    if (!c) return KNH_ERR_INVALID_ARGUMENT;
    if (hipSetDevice(c->device) != hipSuccess) return c->fail(KNH_ERR_DEVICE, "hipSetDevice failed");
    if (hipStreamSynchronize(c->stream) != hipSuccess) return c->fail(KNH_ERR_DEVICE, "hipStreamSynchronize failed");
    c->pending = false;
    return KNH_OK;
  });
}

}  // extern "C"
