// Instantiations of rt_eclipse_qadj (rt_eclipse_qadj.hpp) and their launcher, in a translation unit of their own.
#include "rt_eclipse_qadj.hpp"

#include <cstdlib>

namespace bartrt {

bool launch_rt_qadj(const RtArgs &b, bool sq, int rows, int nblocks, size_t sh, hipStream_t st, hipError_t &err) {
  if (b.A != 5) return false;
#define BARTRT_QADJ(MM, CC)                                                                                       \
  if (b.M == MM && b.C == CC) {                                                                                   \
    if (rows == 16) {                                                                                             \
      if (sq) BARTRT_RT_LAUNCH((rt_eclipse_qadj<5, MM, CC, true, 16>), dim3(nblocks), dim3(256), sh, st, b);      \
      else BARTRT_RT_LAUNCH((rt_eclipse_qadj<5, MM, CC, false, 16>), dim3(nblocks), dim3(256), sh, st, b);        \
    } else {                                                                                                      \
      if (sq) BARTRT_RT_LAUNCH((rt_eclipse_qadj<5, MM, CC, true, 8>), dim3(nblocks), dim3(256), sh, st, b);       \
      else BARTRT_RT_LAUNCH((rt_eclipse_qadj<5, MM, CC, false, 8>), dim3(nblocks), dim3(256), sh, st, b);         \
    }                                                                                                             \
    err = hipGetLastError();                                                                                      \
    return true;                                                                                                  \
  }
  BARTRT_QADJ_LIST(BARTRT_QADJ)
#undef BARTRT_QADJ
  return false;
}

}  // namespace bartrt
