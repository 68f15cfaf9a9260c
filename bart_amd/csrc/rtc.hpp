// Kernels of the hand-written headers instantiated at run time.
//
// The library ships the RT kernels for the shapes BART's examples use, instantiated ahead of time (kernels.hpp,
// BARTRT_MC_LIST: 24 (molecules, cross-section slots) pairs x the five-angle ray grid); every other
// shape -- seven or eight molecules with two cross-section files under the default spline, nine and more molecules, ten
// and more ray angles -- used to fall to the generic kernel, 3-10x slower.  Instead of another 30 MB of objects, such a
// shape is compiled for when it is first launched: the SAME kernel templates (the headers are embedded in the library
// as text, csrc/rtc_sources.inc, written by bart_amd/build.py), instantiated for the exact (angles, molecules, slots,
// rule, cut, rows) by hiprtc, the code object cached in memory and on disk (BARTRT_RTC_CACHE, default
// $XDG_CACHE_HOME/bartrt or ~/.cache/bartrt; keyed by the sources' hash, the kernel and the options).  The
// ahead-of-time set stays as it is: it is what runs where no compiler is at hand (libhiprtc.so missing) or
// BARTRT_RTC=0 is set -- the generic kernel then serves the other shapes as before.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "kernels.hpp"

namespace bartrt {

// is a run-time compiler at hand (and not switched off)?
bool rtc_available();

// Launches bartrt::<expr> (a template-id such as "rt_eclipse_quad<5, 9, 4, true, 8, 1, false, true>") on `args`,
// compiling it first if this process has not seen it.  ilp: under the max-ILP scheduling strategy (the option the
// ahead-of-time build gives the *_ilp translation units).  false: no compiler, or the compilation failed (reported
// once on stderr) -- the caller falls back; err: the launch's status otherwise.
bool rtc_launch(const std::string &expr, bool ilp, dim3 grid, dim3 block, size_t sh, hipStream_t st, const RtArgs &args,
                hipError_t &err);

// compiles bartrt::<expr> and throws the code object away: its size, or -1 with the reason (no GPU needed: the
// embedded sources are held to the compiler by tests/test_capi_cpu.py on machines without one)
long rtc_compile_only(const std::string &expr, bool ilp, std::string &why);

// what the process compiled / loaded so far (bartrt_get_rtc_stats)
struct RtcStats {
  int compiled = 0, from_disk = 0, failed = 0;
  double compile_seconds = 0.0;
};
RtcStats rtc_stats();

}  // namespace bartrt
