// The single-wave eclipse kernels for ONE ray-grid size other than the usual five angles
// (`raygrid` is free-form, examples/demo/BART_eclipse.cfg:135): this file is compiled once per
// angle count (bart_amd/build.py passes -DBARTRT_ANGLES=<n>, n = 1 .. 9 except 5, and the
// max-ILP scheduling option) and instantiates rt_eclipse_fast (rule 0) and rt_eclipse_simpson
// (rule 1), each also in its `cut slant` form (rt_eclipse_fast<..., SLANT>, rt_eclipse_simpson_slant),
// for that count over the (molecules, CIA pairs) list -- without the
// squared-transmittance shortcut, which is tied to the 0 / 60 degree pair of the usual grid.
// launch_rt_spec takes these at every batch size; the quad-layer and producer / consumer
// variants exist for five angles only.  Rule 2 and anything beyond nine angles run the
// generic kernel.
#include "rt_eclipse.hpp"

#ifndef BARTRT_ANGLES
#error "compile with -DBARTRT_ANGLES=<ray-grid size>"
#endif

namespace bartrt {

#define BARTRT_CAT2(a, b) a##b
#define BARTRT_CAT(a, b) BARTRT_CAT2(a, b)

bool BARTRT_CAT(launch_rt_angles_, BARTRT_ANGLES)(const RtArgs &b, int integ, int block, int nblocks, size_t sh,
                                                  hipStream_t st, hipError_t &err) {
  constexpr int A = BARTRT_ANGLES;
#define BARTRT_ANG(MM, CC)                                                                                         \
  if (b.M == MM && b.C == CC) {                                                                                    \
    if (b.cut_slant) {   /* the cut on each ray's slant depth */                                   \
      if (integ == kIntegTransmittance)                                                                            \
        BARTRT_RT_LAUNCH((rt_eclipse_fast<A, MM, CC, false, 0, 1, false, true>), dim3(nblocks), dim3(block), sh, st, b); \
      else                                                                                                         \
        BARTRT_RT_LAUNCH((rt_eclipse_simpson_slant<A, MM, CC, false, (A <= 6 ? 1 : 0)>), dim3(nblocks), dim3(block), sh, st, b); \
    } else if (integ == kIntegTransmittance)                                                                       \
      BARTRT_RT_LAUNCH((rt_eclipse_fast<A, MM, CC, false, 0, 1>), dim3(nblocks), dim3(block), sh, st, b);          \
    else                                                                                                           \
      BARTRT_RT_LAUNCH((rt_eclipse_simpson<A, MM, CC, false, 1>), dim3(nblocks), dim3(block), sh, st, b);          \
    err = hipGetLastError();                                                                                       \
    return true;                                                                                                   \
  }
  BARTRT_MC_LIST(BARTRT_ANG)
#undef BARTRT_ANG
  return false;
}

}  // namespace bartrt
