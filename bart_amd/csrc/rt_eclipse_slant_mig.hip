// rt_eclipse_simpson_slant<..., MIG> (rt_eclipse_s1s.hpp): rule 1's single-wave `cut slant` kernel whose columns
// migrate between SIMDs, for the shapes of BARTRT_QADJ_LIST (others: instantiated at run time, rtc.hpp); compiled under
// the maximum-ILP scheduling strategy like the kernel it is a form of (bart_amd/build.py).
#include "rt_eclipse.hpp"

namespace bartrt {

bool launch_rt_slant_mig(const RtArgs &b, bool sq, int nblocks, size_t sh, hipStream_t st, hipError_t &err) {
#define BARTRT_SLANT_MIG(MM, CC)                                                                                            \
  if (b.M == MM && b.C == CC) {                                                                                             \
    if (sq) BARTRT_RT_LAUNCH((rt_eclipse_simpson_slant<5, MM, CC, true, 1, false, false, true>), dim3(nblocks), dim3(64), sh, st, b);  \
    else BARTRT_RT_LAUNCH((rt_eclipse_simpson_slant<5, MM, CC, false, 1, false, false, true>), dim3(nblocks), dim3(64), sh, st, b);    \
    err = hipGetLastError();                                                                                                \
    return true;                                                                                                            \
  }
  BARTRT_QADJ_LIST(BARTRT_SLANT_MIG)
#undef BARTRT_SLANT_MIG
  return false;
}

}  // namespace bartrt
