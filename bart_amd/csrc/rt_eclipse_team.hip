// The three-wave column team of `cut slant`, rule 1, five ray angles (rt_eclipse_s1t.hpp): launch_rt_spec
// takes it between the few-(walker, wavenumber) range of the one-ray-per-lane kernel and the batch sizes at
// which single-wave columns fill the chip evenly by themselves.
#include "rt_eclipse.hpp"

namespace bartrt {

bool launch_rt_slant_team(const RtArgs &b, bool sq, int nblocks, size_t sh, hipStream_t st, hipError_t &err) {
#define BARTRT_TEAM(MM, CC)                                                                              \
  if (b.M == MM && b.C == CC) {                                                                          \
    if (sq) BARTRT_RT_LAUNCH((rt_eclipse_slant_team<MM, CC, true>), dim3(nblocks), dim3(192), sh, st, b);  \
    else BARTRT_RT_LAUNCH((rt_eclipse_slant_team<MM, CC, false>), dim3(nblocks), dim3(192), sh, st, b);    \
    err = hipGetLastError();                                                                             \
    return true;                                                                                         \
  }
  BARTRT_MC_LIST(BARTRT_TEAM)
#undef BARTRT_TEAM
  return false;
}

}  // namespace bartrt
