// The single-wave eclipse kernel of integration rule 0, compiled under the
// compiler's maximum-ILP scheduling strategy (bart_amd/build.py passes
// -mllvm -amdgpu-sched-strategy=max-ilp for this file only): see rt_eclipse_fast
// in rt_eclipse.hpp for what that changes and when launch_rt_spec takes this build.
#include "rt_eclipse.hpp"

namespace bartrt {

bool launch_rt_fast_ilp(const RtArgs &b, bool sq, int block, int nblocks, size_t sh, hipStream_t st, hipError_t &err) {
#define BARTRT_FAST_ILP(MM, CC)                                                                                    \
  if (b.M == MM && b.C == CC) {                                                                                    \
    if (sq) BARTRT_RT_LAUNCH((rt_eclipse_fast<5, MM, CC, true, 0, 1>), dim3(nblocks), dim3(block), sh, st, b);   \
    else BARTRT_RT_LAUNCH((rt_eclipse_fast<5, MM, CC, false, 0, 1>), dim3(nblocks), dim3(block), sh, st, b);     \
    err = hipGetLastError();                                                                                       \
    return true;                                                                                                   \
  }
  BARTRT_MC_LIST(BARTRT_FAST_ILP)
#undef BARTRT_FAST_ILP
  return false;
}

bool launch_rt_fast_ext(const RtArgs &b, bool sq, int block, int nblocks, size_t sh, hipStream_t st, hipError_t &err) {
#define BARTRT_FAST_EXT(CC)                                                                                           \
  if (b.M == 0 && b.C == CC) {                                                                                        \
    if (sq) BARTRT_RT_LAUNCH((rt_eclipse_fast<5, 0, CC, true, 0, 1, true>), dim3(nblocks), dim3(block), sh, st, b);  \
    else BARTRT_RT_LAUNCH((rt_eclipse_fast<5, 0, CC, false, 0, 1, true>), dim3(nblocks), dim3(block), sh, st, b);    \
    err = hipGetLastError();                                                                                          \
    return true;                                                                                                      \
  }
  BARTRT_EXT_C_LIST(BARTRT_FAST_EXT)
#undef BARTRT_FAST_EXT
  return false;
}

}  // namespace bartrt

// include/bartrt.h: what the measured table names for a launch (no GPU, no engine)
extern "C" const char *bartrt_kernel_choice(int nmol, long columns) {
  return bartrt::kernel_variant_name(bartrt::slant_simpson_choice(nmol, columns < 0 ? 0 : columns).variant);
}
