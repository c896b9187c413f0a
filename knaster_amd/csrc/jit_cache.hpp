// jit_cache.hpp -- run-time fusion outside the host's blast radius (host code only; no HIP types).
//
// Two things, used by jit.hip and by the helper program jit_helper.cpp:
//   * a code-object cache on disk: one file per kernel identity, named by the SHA-256 of everything the generated code
//     depends on (the device source text, the kernel's name expression, the compiler options, the hiprtc version this
//     library was built against and the one it runs with), so that a second process -- or a second knh_bank_init -- loads
//     in milliseconds what the first one compiled in seconds or minutes;
//   * the compile itself in a helper PROCESS (knh_jit_helper, found beside the library): hiprtc is a whole compiler, and
//     a compiler that crashes (it has: DESIGN.md section 4) takes its process with it.  In the helper that is a status and a
//     message (KNH_ERR_INTERNAL + knh_last_error) for the host, which the reference's "never abort on the audio path" rule
//     (Errors) asks for; hiprtc needs no device, and the helper never touches one.
//
// Environment: KNH_JIT_CACHE_DIR (default $XDG_CACHE_HOME/knaster_hip, $HOME/.cache/knaster_hip, /tmp/knaster_hip-<uid>),
// KNH_JIT_CACHE=0 (no disk cache), KNH_JIT_HELPER=<path> (the helper program), KNH_JIT_INPROCESS=1 (compile in the host
// process, as rounds 1-3 did), KNH_JIT_TIMEOUT_S (default 1800: a 200-stage voice takes 100 s).
#pragma once
#include <fcntl.h>
#include <signal.h>
#include <spawn.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cerrno>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

extern char** environ;

namespace knh_jit {

// ---- SHA-256 (FIPS 180-4) ------------------------------------------------------------------------------------------
struct Sha256 {
  uint32_t h[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
  uint8_t buf[64];
  size_t fill = 0;
  uint64_t total = 0;
  static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
  void block(const uint8_t* p) {
    static const uint32_t k[64] = {
        0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u, 0x12835b01u, 0x243185beu,
        0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau,
        0x5cb0a9dcu, 0x76f988dau, 0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u, 0x27b70a85u,
        0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u, 0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u,
        0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u, 0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu,
        0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
    uint32_t w[64];
    for (int i = 0; i < 16; ++i) w[i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | p[4 * i + 3];
    for (int i = 16; i < 64; ++i) {
      const uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
      w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; ++i) {
      const uint32_t t1 = hh + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + k[i] + w[i];
      const uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
      hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
  }
  void update(const void* data, size_t n) {
    const uint8_t* p = static_cast<const uint8_t*>(data);
    total += n;
    while (n > 0) {
      const size_t take = std::min(n, sizeof(buf) - fill);
      std::memcpy(buf + fill, p, take);
      fill += take; p += take; n -= take;
      if (fill == 64) { block(buf); fill = 0; }
    }
  }
  void field(const std::string& s) {  // length-prefixed: ("ab", "c") and ("a", "bc") hash differently
    const uint64_t n = s.size();
    update(&n, sizeof n);
    update(s.data(), s.size());
  }
  std::string hex() {
    const uint64_t bits = total * 8;
    const uint8_t one = 0x80, zero = 0;
    update(&one, 1);
    while (fill != 56) update(&zero, 1);
    uint8_t len[8];
    for (int i = 0; i < 8; ++i) len[i] = (uint8_t)(bits >> (56 - 8 * i));
    update(len, 8);
    char out[65];
    for (int i = 0; i < 8; ++i) std::snprintf(out + 8 * i, 9, "%08x", h[i]);
    return std::string(out, 64);
  }
};

// ---- what is compiled ------------------------------------------------------------------------------------------------
struct Job {
  std::string source;              // the whole translation unit
  std::string file_name;           // the name hiprtc reports it under
  std::string name_expression;     // a template instantiation to lower ("" : the kernel is extern "C")
  std::string fixed_lowered_name;  // the kernel's name when there is no name expression
  std::vector<std::string> options;
};
struct Code {
  std::vector<char> object;
  std::string lowered_name;
};
inline std::string digest(const Job& j, const std::string& toolchain) {
  Sha256 s;
  s.field("knaster_hip code object v1");
  s.field(toolchain);
  s.field(j.file_name);
  s.field(j.name_expression);
  s.field(j.fixed_lowered_name);
  for (const std::string& o : j.options) s.field(o);
  s.field(j.source);
  return s.hex();
}

// ---- files -----------------------------------------------------------------------------------------------------------
inline bool read_file(const std::string& path, std::string* out) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) return false;
  std::string data;
  char chunk[1 << 16];
  size_t n;
  while ((n = std::fread(chunk, 1, sizeof chunk, f)) > 0) data.append(chunk, n);
  const bool ok = !std::ferror(f);
  std::fclose(f);
  if (ok) out->swap(data);
  return ok;
}
// write to a temporary name in the same directory, then rename: a reader sees the whole file or none of it
inline bool write_file_atomic(const std::string& path, const std::string& data) {
  const std::string tmp = path + ".tmp." + std::to_string((long)getpid()) + "." + std::to_string((long)std::chrono::steady_clock::now().time_since_epoch().count());
  FILE* f = std::fopen(tmp.c_str(), "wb");
  if (!f) return false;
  const bool ok = std::fwrite(data.data(), 1, data.size(), f) == data.size() && std::fflush(f) == 0;
  std::fclose(f);
  if (!ok || std::rename(tmp.c_str(), path.c_str()) != 0) { std::remove(tmp.c_str()); return false; }
  return true;
}
inline void put_u64(std::string* s, uint64_t v) { s->append(reinterpret_cast<const char*>(&v), 8); }
inline bool get_u64(const std::string& s, size_t* pos, uint64_t* v) {
  if (*pos + 8 > s.size()) return false;
  std::memcpy(v, s.data() + *pos, 8);
  *pos += 8;
  return true;
}
inline bool get_str(const std::string& s, size_t* pos, std::string* out) {
  uint64_t n = 0;
  if (!get_u64(s, pos, &n) || n > s.size() - *pos) return false;
  out->assign(s, *pos, n);
  *pos += n;
  return true;
}
inline void put_str(std::string* s, const std::string& v) { put_u64(s, v.size()); s->append(v); }

// cache entry: magic | digest | lowered name | code object | SHA-256 of the three (a torn or tampered file is a miss)
constexpr const char* kMagic = "KNHCO001";
inline std::string encode_entry(const std::string& dig, const Code& c) {
  std::string body;
  put_str(&body, dig);
  put_str(&body, c.lowered_name);
  put_str(&body, std::string(c.object.data(), c.object.size()));
  Sha256 s;
  s.update(body.data(), body.size());
  return std::string(kMagic, 8) + body + s.hex();
}
inline bool decode_entry(const std::string& data, const std::string& dig, Code* out) {
  if (data.size() < 8 + 64 || data.compare(0, 8, kMagic, 8) != 0) return false;
  const std::string body = data.substr(8, data.size() - 8 - 64);
  Sha256 s;
  s.update(body.data(), body.size());
  if (s.hex() != data.substr(data.size() - 64)) return false;
  size_t pos = 0;
  std::string d, lowered, obj;
  if (!get_str(body, &pos, &d) || !get_str(body, &pos, &lowered) || !get_str(body, &pos, &obj) || pos != body.size() || d != dig) return false;
  out->lowered_name = lowered;
  out->object.assign(obj.begin(), obj.end());
  return true;
}
inline std::string encode_job(const Job& j) {
  std::string s = "KNHJOB01";
  put_str(&s, j.source);
  put_str(&s, j.file_name);
  put_str(&s, j.name_expression);
  put_str(&s, j.fixed_lowered_name);
  put_u64(&s, j.options.size());
  for (const std::string& o : j.options) put_str(&s, o);
  return s;
}
inline bool decode_job(const std::string& data, Job* j) {
  if (data.compare(0, 8, "KNHJOB01", 8) != 0) return false;
  size_t pos = 8;
  uint64_t n = 0;
  if (!get_str(data, &pos, &j->source) || !get_str(data, &pos, &j->file_name) || !get_str(data, &pos, &j->name_expression) ||
      !get_str(data, &pos, &j->fixed_lowered_name) || !get_u64(data, &pos, &n) || n > 64)
    return false;
  j->options.resize(n);
  for (std::string& o : j->options)
    if (!get_str(data, &pos, &o)) return false;
  return pos == data.size();
}

inline bool make_dirs(const std::string& path) {
  for (size_t i = 1; i <= path.size(); ++i) {
    if (i == path.size() || path[i] == '/') {
      const std::string part = path.substr(0, i);
      if (mkdir(part.c_str(), 0700) != 0 && errno != EEXIST) return false;
    }
  }
  struct stat st{};
  return stat(path.c_str(), &st) == 0 && S_ISDIR(st.st_mode) && access(path.c_str(), W_OK | X_OK) == 0;
}
// "" when there is to be no disk cache (KNH_JIT_CACHE=0) or no directory can be had
inline std::string cache_dir() {
  const char* off = std::getenv("KNH_JIT_CACHE");
  if (off && off[0] == '0') return "";
  std::vector<std::string> candidates;
  if (const char* d = std::getenv("KNH_JIT_CACHE_DIR")) { if (d[0]) candidates.push_back(d); }
  else {
    if (const char* x = std::getenv("XDG_CACHE_HOME")) if (x[0] == '/') candidates.push_back(std::string(x) + "/knaster_hip");
    if (const char* h = std::getenv("HOME")) if (h[0] == '/') candidates.push_back(std::string(h) + "/.cache/knaster_hip");
    candidates.push_back("/tmp/knaster_hip-" + std::to_string((long)getuid()));
  }
  for (const std::string& c : candidates)
    if (make_dirs(c)) return c;
  return "";
}

// ---- the helper process ----------------------------------------------------------------------------------------------
// Runs `helper job_path out_path log_path digest` and waits for it.  Returns 0 and leaves the entry at out_path; or a negative
// number with *error set: -1 the helper could not be started, -2 it died (a signal: the compiler crashed), -3 it timed out and
// was killed, -4 it reported a compile error (exit status 1; *error holds the compiler's log).
inline int run_helper(const std::string& helper, const std::string& job_path, const std::string& out_path, const std::string& log_path,
                      const std::string& dig, double timeout_s, std::string* error) {
  std::vector<char*> argv = {const_cast<char*>(helper.c_str()), const_cast<char*>(job_path.c_str()), const_cast<char*>(out_path.c_str()),
                             const_cast<char*>(log_path.c_str()), const_cast<char*>(dig.c_str()), nullptr};
  posix_spawn_file_actions_t fa;
  posix_spawn_file_actions_init(&fa);
  posix_spawn_file_actions_addopen(&fa, 0, "/dev/null", O_RDONLY, 0);
  posix_spawnattr_t at;
  posix_spawnattr_init(&at);
  sigset_t none;
  sigemptyset(&none);
  posix_spawnattr_setsigmask(&at, &none);  // (the host may block signals on its audio thread; the compiler should not inherit that)
  posix_spawnattr_setflags(&at, POSIX_SPAWN_SETSIGMASK);
  pid_t pid = 0;
  const int rc = posix_spawn(&pid, helper.c_str(), &fa, &at, argv.data(), environ);
  posix_spawn_file_actions_destroy(&fa);
  posix_spawnattr_destroy(&at);
  if (rc != 0) { *error = "could not start the JIT helper " + helper + ": " + std::strerror(rc); return -1; }
  const auto t0 = std::chrono::steady_clock::now();
  int status = 0;
  for (unsigned spin = 0;; ++spin) {
    const pid_t w = waitpid(pid, &status, WNOHANG);
    if (w == pid) break;
    if (w < 0 && errno != EINTR) { *error = std::string("waitpid for the JIT helper: ") + std::strerror(errno); return -1; }
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) {
      kill(pid, SIGKILL);
      (void)waitpid(pid, &status, 0);
      *error = "the JIT helper did not finish within " + std::to_string((long)timeout_s) + " s and was killed (KNH_JIT_TIMEOUT_S)";
      return -3;
    }
    std::this_thread::sleep_for(std::chrono::microseconds(spin < 200 ? 500 : 5000));
  }
  std::string log;
  (void)read_file(log_path, &log);
  if (log.size() > 3000) log.resize(3000);
  if (WIFSIGNALED(status)) {
    *error = "the compiler crashed in the JIT helper process (signal " + std::to_string(WTERMSIG(status)) + "); the host is unharmed" + (log.empty() ? "" : "\n" + log);
    return -2;
  }
  if (!WIFEXITED(status) || WEXITSTATUS(status) != 0) {
    *error = log.empty() ? "the JIT helper failed with exit status " + std::to_string(WIFEXITED(status) ? WEXITSTATUS(status) : -1) : log;
    return WIFEXITED(status) && WEXITSTATUS(status) == 1 ? -4 : -2;
  }
  return 0;
}

}  // namespace knh_jit
