// Run-time instantiation of the RT kernels: see rtc.hpp.
#include "rtc.hpp"

#include <hip/hiprtc.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <mutex>
#include <vector>

#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include "rtc_sources.inc"   // kRtcHeaderNames[], kRtcHeaderSources[], kRtcNumHeaders, kRtcSourceId (bart_amd/build.py)

namespace bartrt {

namespace {

struct Api {
  void *lib = nullptr;
  decltype(&hiprtcCreateProgram) create = nullptr;
  decltype(&hiprtcDestroyProgram) destroy = nullptr;
  decltype(&hiprtcAddNameExpression) add_name = nullptr;
  decltype(&hiprtcCompileProgram) compile = nullptr;
  decltype(&hiprtcGetProgramLogSize) log_size = nullptr;
  decltype(&hiprtcGetProgramLog) log = nullptr;
  decltype(&hiprtcGetLoweredName) lowered = nullptr;
  decltype(&hiprtcGetCodeSize) code_size = nullptr;
  decltype(&hiprtcGetCode) code = nullptr;
  decltype(&hiprtcVersion) version = nullptr;
  int vmajor = 0, vminor = 0;      // of the compiler at hand: part of the disk cache's key
  bool ok = false;
};

Api &api() {
  static Api a = [] {
    Api x;
    const char *off = std::getenv("BARTRT_RTC");
    if (off && off[0] == '0') return x;
    for (const char *name : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"}) {
      x.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (x.lib) break;
    }
    if (!x.lib) return x;
#define BARTRT_SYM(field, sym) x.field = reinterpret_cast<decltype(x.field)>(dlsym(x.lib, #sym))
    BARTRT_SYM(create, hiprtcCreateProgram); BARTRT_SYM(destroy, hiprtcDestroyProgram);
    BARTRT_SYM(add_name, hiprtcAddNameExpression); BARTRT_SYM(compile, hiprtcCompileProgram);
    BARTRT_SYM(log_size, hiprtcGetProgramLogSize); BARTRT_SYM(log, hiprtcGetProgramLog);
    BARTRT_SYM(lowered, hiprtcGetLoweredName); BARTRT_SYM(code_size, hiprtcGetCodeSize); BARTRT_SYM(code, hiprtcGetCode);
#undef BARTRT_SYM
    x.version = reinterpret_cast<decltype(x.version)>(dlsym(x.lib, "hiprtcVersion"));
    if (x.version) (void)x.version(&x.vmajor, &x.vminor);
    x.ok = x.create && x.destroy && x.add_name && x.compile && x.log_size && x.log && x.lowered && x.code_size && x.code;
    return x;
  }();
  return a;
}

struct Entry { hipFunction_t fn = nullptr; bool failed = false; };
std::mutex g_mu;
std::map<std::string, Entry> g_cache;
std::vector<hipModule_t> g_modules;
RtcStats g_stats;

// The directory of the on-disk cache, or "" when there is none to trust: a code object read from it is loaded and run
// on the GPU against the engine's buffers, so the directory must be this user's own and writable by nobody else (a
// /tmp fallback or a shared BARTRT_RTC_CACHE another local user pre-created does not qualify: the process then compiles
// in memory only).
std::string cache_dir() {
  static const std::string dir = [] {
    std::string d;
    if (const char *e = std::getenv("BARTRT_RTC_CACHE")) d = e;
    else if (const char *x = std::getenv("XDG_CACHE_HOME")) d = std::string(x) + "/bartrt";
    else if (const char *h = std::getenv("HOME")) d = std::string(h) + "/.cache/bartrt";
    else d = "/tmp/bartrt_cache_" + std::to_string((long)getuid());
    // (mkdir -p, two levels are enough for the defaults)
    const size_t slash = d.find_last_of('/');
    if (slash != std::string::npos && slash > 0) (void)mkdir(d.substr(0, slash).c_str(), 0700);
    (void)mkdir(d.c_str(), 0700);
    struct stat st;
    if (lstat(d.c_str(), &st) != 0 || !S_ISDIR(st.st_mode) || st.st_uid != getuid() || (st.st_mode & (S_IWGRP | S_IWOTH))) {
      std::fprintf(stderr, "libbartrt: kernel cache directory %s is not a directory of this user's own without group/other "
                           "write permission: compiled kernels are kept in memory only\n", d.c_str());
      return std::string();
    }
    return d;
  }();
  return dir;
}

std::string hex64(const std::string &s) {
  unsigned long long h = 1469598103934665603ull;
  for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; }
  char b[24];
  std::snprintf(b, sizeof b, "%016llx", h);
  return b;
}

// cache file: "BARTRTC1" | u32 name length | lowered name | code object
bool read_cached(const std::string &path, std::string &lowered, std::vector<char> &code) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  char magic[8];
  unsigned n = 0;
  if (!f.read(magic, 8) || std::memcmp(magic, "BARTRTC1", 8) != 0 || !f.read(reinterpret_cast<char *>(&n), 4) || n == 0 || n > 4096) return false;
  lowered.resize(n);
  if (!f.read(&lowered[0], n)) return false;
  code.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
  return code.size() > 64;
}

void write_cached(const std::string &path, const std::string &lowered, const std::vector<char> &code) {
  const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
  {
    std::ofstream f(tmp, std::ios::binary);
    if (!f) return;
    const unsigned n = (unsigned)lowered.size();
    f.write("BARTRTC1", 8);
    f.write(reinterpret_cast<const char *>(&n), 4);
    f.write(lowered.data(), n);
    f.write(code.data(), (std::streamsize)code.size());
    if (!f) { (void)unlink(tmp.c_str()); return; }
  }
  if (rename(tmp.c_str(), path.c_str()) != 0) (void)unlink(tmp.c_str());
}

bool compile(const std::string &expr, bool ilp, std::string &lowered, std::vector<char> &code, std::string &why) {
  Api &a = api();
  const std::string full = "bartrt::" + expr;
  // one translation unit: the kernel headers, one explicit name
  const std::string src = "#include \"rt_eclipse.hpp\"\n#include \"rt_eclipse_qadj.hpp\"\n";
  hiprtcProgram prog = nullptr;
  if (a.create(&prog, src.c_str(), "bartrt_rtc.hip", kRtcNumHeaders, kRtcHeaderSources, kRtcHeaderNames) != HIPRTC_SUCCESS) {
    why = "hiprtcCreateProgram failed";
    return false;
  }
  bool ok = false;
  do {
    if (a.add_name(prog, full.c_str()) != HIPRTC_SUCCESS) { why = "hiprtcAddNameExpression failed"; break; }
    std::vector<const char *> opts = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-result"};
    if (ilp) { opts.push_back("-mllvm"); opts.push_back("-amdgpu-sched-strategy=max-ilp"); }
    const hiprtcResult r = a.compile(prog, (int)opts.size(), opts.data());
    if (r != HIPRTC_SUCCESS) {
      size_t n = 0;
      a.log_size(prog, &n);
      std::string log(n, '\0');
      if (n) a.log(prog, &log[0]);
      why = "hiprtc could not compile " + full + ":\n" + log.substr(0, 4000);
      break;
    }
    const char *low = nullptr;
    if (a.lowered(prog, full.c_str(), &low) != HIPRTC_SUCCESS || !low) { why = "no lowered name for " + full; break; }
    lowered = low;
    size_t cs = 0;
    if (a.code_size(prog, &cs) != HIPRTC_SUCCESS || cs == 0) { why = "empty code object for " + full; break; }
    code.resize(cs);
    if (a.code(prog, code.data()) != HIPRTC_SUCCESS) { why = "hiprtcGetCode failed"; break; }
    ok = true;
  } while (false);
  a.destroy(&prog);
  return ok;
}

hipFunction_t get(const std::string &expr, bool ilp) {
  const std::string key = expr + (ilp ? "|ilp" : "|occ");
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_cache.find(key);
  if (it != g_cache.end()) return it->second.failed ? nullptr : it->second.fn;
  Entry e;
  e.failed = true;
  std::string lowered;
  std::vector<char> code;
  const std::string cdir = cache_dir();
  const std::string path = cdir.empty() ? std::string() : cdir + "/" + hex64(std::string(kRtcSourceId) + "|gfx950|hiprtc " + std::to_string(api().vmajor) + "." +
                                                     std::to_string(api().vminor) + "|" + key) + ".hsaco";
  bool have = !path.empty() && read_cached(path, lowered, code);
  if (have) {
    g_stats.from_disk++;
  } else if (api().ok) {
    const auto t0 = std::chrono::steady_clock::now();
    std::string why;
    have = compile(expr, ilp, lowered, code, why);
    g_stats.compile_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (have) {
      g_stats.compiled++;
      if (!path.empty()) write_cached(path, lowered, code);
      if (std::getenv("BARTRT_RTC_VERBOSE")) std::fprintf(stderr, "libbartrt: compiled %s (%zu bytes)\n", expr.c_str(), code.size());
    } else {
      g_stats.failed++;
      std::fprintf(stderr, "libbartrt: %s\n(falling back to the generic kernel for this shape)\n", why.c_str());
    }
  }
  if (have) {
    hipModule_t mod = nullptr;
    if (hipModuleLoadData(&mod, code.data()) == hipSuccess && hipModuleGetFunction(&e.fn, mod, lowered.c_str()) == hipSuccess) {
      g_modules.push_back(mod);
      e.failed = false;
      // (layer records above the 64 kB default of dynamic LDS: the specialised kernels' launchers cap at 55 kB)
    } else {
      (void)hipGetLastError();
      g_stats.failed++;
      std::fprintf(stderr, "libbartrt: could not load the compiled kernel %s\n", expr.c_str());
    }
  }
  g_cache[key] = e;
  return e.failed ? nullptr : e.fn;
}

}  // namespace

bool rtc_available() { return api().ok; }

long rtc_compile_only(const std::string &expr, bool ilp, std::string &why) {
  if (!api().ok) { why = "no run-time compiler (libhiprtc.so not found, or BARTRT_RTC=0)"; return -1; }
  std::string lowered;
  std::vector<char> code;
  if (!compile(expr, ilp, lowered, code, why)) return -1;
  return (long)code.size();
}

RtcStats rtc_stats() {
  std::lock_guard<std::mutex> lock(g_mu);
  return g_stats;
}

bool rtc_launch(const std::string &expr, bool ilp, dim3 grid, dim3 block, size_t sh, hipStream_t st, const RtArgs &args,
                hipError_t &err) {
  static const bool off = [] { const char *e = std::getenv("BARTRT_RTC"); return e && e[0] == '0'; }();
  if (off) return false;
  // (cached code objects load without a compiler; only a miss needs hiprtc)
  hipFunction_t fn = get(expr, ilp);
  if (!fn) return false;
  RtArgs a = args;
  size_t size = sizeof(RtArgs);
  void *config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
  if (args.ev_start || args.ev_stop) {
    err = hipExtModuleLaunchKernel(fn, grid.x * block.x, grid.y * block.y, grid.z * block.z, block.x, block.y, block.z, sh, st,
                                   nullptr, config, args.ev_start, args.ev_stop, 0);
  } else {
    err = hipModuleLaunchKernel(fn, grid.x, grid.y, grid.z, block.x, block.y, block.z, (unsigned)sh, st, nullptr, config);
  }
  return true;
}

}  // namespace bartrt
