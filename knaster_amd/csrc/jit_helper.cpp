// jit_helper.cpp -> knh_jit_helper: the process that runs hiprtc for libknaster_hip.so (jit_cache.hpp).
//   knh_jit_helper <job file> <output file> <log file> <digest>
// Reads one compile job, compiles it for gfx950 and writes the cache entry (digest | lowered name | code object); exit
// status 0 = done, 1 = the compiler refused the program (its log is in the log file), anything else or a signal = the
// compiler crashed.  It links hiprtc only and never touches a device: the host process keeps the GPU, this one only the
// compiler -- and whatever the compiler does to its process, it does to this one.
// Built by knaster_amd/build.py with g++ (host code): -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -lhiprtc.
#include <hip/hiprtc.h>

#include "jit_cache.hpp"

static int fail(const std::string& log_path, const std::string& msg, int code) {
  (void)knh_jit::write_file_atomic(log_path, msg);
  return code;
}

int main(int argc, char** argv) {
  if (argc != 5) { std::fprintf(stderr, "usage: knh_jit_helper <job> <out> <log> <digest>   (started by libknaster_hip.so)\n"); return 2; }
  const std::string job_path = argv[1], out_path = argv[2], log_path = argv[3], dig = argv[4];
  std::string data;
  knh_jit::Job job;
  if (!knh_jit::read_file(job_path, &data) || !knh_jit::decode_job(data, &job)) return fail(log_path, "knh_jit_helper: unreadable job file " + job_path, 3);
  // test hooks: what a crashing / hanging compiler does to the host (tests/test_jit_cache.py)
  if (const char* t = std::getenv("KNH_JIT_HELPER_TEST")) {
    if (!std::strcmp(t, "crash")) std::abort();
    if (!std::strcmp(t, "segv")) { volatile int* p = nullptr; *p = 1; }
    if (!std::strcmp(t, "hang")) for (;;) pause();
  }
  hiprtcProgram prog = nullptr;
  auto rtc_fail = [&](const std::string& what, hiprtcResult r) {
    std::string msg = what + ": " + hiprtcGetErrorString(r);
    if (prog) {
      size_t n = 0;
      if (hiprtcGetProgramLogSize(prog, &n) == HIPRTC_SUCCESS && n > 1) {
        std::string log(n, '\0');
        if (hiprtcGetProgramLog(prog, &log[0]) == HIPRTC_SUCCESS) msg += "\n" + log.substr(0, 2000);
      }
      hiprtcDestroyProgram(&prog);
    }
    return fail(log_path, msg, 1);
  };
  hiprtcResult r = hiprtcCreateProgram(&prog, job.source.c_str(), job.file_name.c_str(), 0, nullptr, nullptr);
  if (r != HIPRTC_SUCCESS) return rtc_fail("hiprtcCreateProgram", r);
  if (!job.name_expression.empty()) {
    r = hiprtcAddNameExpression(prog, job.name_expression.c_str());
    if (r != HIPRTC_SUCCESS) return rtc_fail("hiprtcAddNameExpression", r);
  }
  std::vector<const char*> opts;
  for (const std::string& o : job.options) opts.push_back(o.c_str());
  r = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
  if (r != HIPRTC_SUCCESS) return rtc_fail("hiprtcCompileProgram(" + (job.name_expression.empty() ? job.fixed_lowered_name : job.name_expression) + ")", r);
  knh_jit::Code code;
  code.lowered_name = job.fixed_lowered_name;
  if (!job.name_expression.empty()) {
    const char* lowered = nullptr;
    r = hiprtcGetLoweredName(prog, job.name_expression.c_str(), &lowered);
    if (r != HIPRTC_SUCCESS) return rtc_fail("hiprtcGetLoweredName", r);
    code.lowered_name = lowered;
  }
  size_t n = 0;
  r = hiprtcGetCodeSize(prog, &n);
  if (r != HIPRTC_SUCCESS) return rtc_fail("hiprtcGetCodeSize", r);
  code.object.resize(n);
  r = hiprtcGetCode(prog, code.object.data());
  if (r != HIPRTC_SUCCESS) return rtc_fail("hiprtcGetCode", r);
  hiprtcDestroyProgram(&prog);
  // (the digest is the host's business: it names the entry and is checked when the entry is read back)
  if (!knh_jit::write_file_atomic(out_path, knh_jit::encode_entry(dig, code))) return fail(log_path, "knh_jit_helper: cannot write " + out_path, 3);
  return 0;
}
