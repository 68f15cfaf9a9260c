// The single-wave eclipse kernels with the `toomuch` cut on each ray's slant depth (cfg `cut slant`,
// DESIGN.md C19) for the usual five-angle ray grid, compiled under the compiler's maximum-ILP
// scheduling strategy (bart_amd/build.py): rule 1's own kernel (rt_eclipse_simpson_slant,
// rt_eclipse_s1s.hpp), rule 0's (rt_eclipse_fast<..., SLANT = true> with ColumnFluxSlant, integ.hpp),
// rule 2's (the same kernel with rule 2's masked accumulator), and rules 0 / 1 with the line-by-line extinction
// array as input.
#include "rt_eclipse.hpp"

// (A/B builds, tools/ab_build.py: the single-wave slant kernel with or without the record read-ahead)
#ifndef BARTRT_SLANT_SCHED
#define BARTRT_SLANT_SCHED 1
#endif

namespace bartrt {

// The record read-ahead (SCHED 1: a layer reads the NEXT layer's record while it computes) pays on the shapes it was
// tuned on -- up to twelve table loads per layer -- and drowns wider ones in spills: <5, 6, 2> (sixteen loads) 294
// registers spilled against 22 without it, <5, 4, 4> (BART's usual H2-H2 + H2-He under the spline) 161 against 26;
// measured on six molecules: 762 -> 3xx us at 26 walkers (round 6, slant_sched in rt_eclipse.hpp).

bool launch_rt_slant(const RtArgs &b, int integ, bool sq, int block, int nblocks, size_t sh, hipStream_t st, hipError_t &err) {
#define BARTRT_SLANT(MM, CC)                                                                                              \
  if (b.M == MM && b.C == CC) {                                                                                           \
    if (integ == kIntegSimpson) {                                                                                         \
      if (sq) BARTRT_RT_LAUNCH((rt_eclipse_simpson_slant<5, MM, CC, true, slant_sched(MM, CC, BARTRT_SLANT_SCHED)>), dim3(nblocks), dim3(block), sh, st, b);    \
      else BARTRT_RT_LAUNCH((rt_eclipse_simpson_slant<5, MM, CC, false, slant_sched(MM, CC, BARTRT_SLANT_SCHED)>), dim3(nblocks), dim3(block), sh, st, b);      \
    } else if (integ == kIntegTransmittance) {                                                                            \
      if (sq) BARTRT_RT_LAUNCH((rt_eclipse_fast<5, MM, CC, true, 0, 1, false, true>), dim3(nblocks), dim3(block), sh, st, b);  \
      else BARTRT_RT_LAUNCH((rt_eclipse_fast<5, MM, CC, false, 0, 1, false, true>), dim3(nblocks), dim3(block), sh, st, b);    \
    } else {                                                                                                              \
      if (sq) BARTRT_RT_LAUNCH((rt_eclipse_fast<5, MM, CC, true, 2, 1, false, true>), dim3(nblocks), dim3(block), sh, st, b);  \
      else BARTRT_RT_LAUNCH((rt_eclipse_fast<5, MM, CC, false, 2, 1, false, true>), dim3(nblocks), dim3(block), sh, st, b);    \
    }                                                                                                                     \
    err = hipGetLastError();                                                                                              \
    return true;                                                                                                          \
  }
  BARTRT_MC_LIST(BARTRT_SLANT)
#undef BARTRT_SLANT
  return false;
}

// rule 1 with the optical-depth / per-ray-intensity outputs of a single walker (rt_eclipse_simpson_slant<..., OUT>;
// the ray angles in the cfg's order: no squared-transmittance pairing)
bool launch_rt_slant_out(const RtArgs &b, int block, int nblocks, size_t sh, hipStream_t st, hipError_t &err) {
#define BARTRT_SLANT_OUT(MM, CC)                                                                                          \
  if (b.M == MM && b.C == CC) {                                                                                           \
    BARTRT_RT_LAUNCH((rt_eclipse_simpson_slant<5, MM, CC, false, 0, false, true>), dim3(nblocks), dim3(block), sh, st, b); \
    err = hipGetLastError();                                                                                              \
    return true;                                                                                                          \
  }
  BARTRT_MC_LIST(BARTRT_SLANT_OUT)
#undef BARTRT_SLANT_OUT
  return false;
}

bool launch_rt_slant_ext(const RtArgs &b, int integ, bool sq, int block, int nblocks, size_t sh, hipStream_t st, hipError_t &err) {
  if (integ != kIntegSimpson && integ != kIntegTransmittance) return false;
#define BARTRT_SLANT_EXT(CC)                                                                                                   \
  if (b.M == 0 && b.C == CC) {                                                                                                 \
    if (integ == kIntegSimpson) {                                                                                              \
      if (sq) BARTRT_RT_LAUNCH((rt_eclipse_simpson_slant<5, 0, CC, true, 1, true>), dim3(nblocks), dim3(block), sh, st, b);    \
      else BARTRT_RT_LAUNCH((rt_eclipse_simpson_slant<5, 0, CC, false, 1, true>), dim3(nblocks), dim3(block), sh, st, b);      \
    } else {                                                                                                                   \
      if (sq) BARTRT_RT_LAUNCH((rt_eclipse_fast<5, 0, CC, true, 0, 1, true, true>), dim3(nblocks), dim3(block), sh, st, b);    \
      else BARTRT_RT_LAUNCH((rt_eclipse_fast<5, 0, CC, false, 0, 1, true, true>), dim3(nblocks), dim3(block), sh, st, b);      \
    }                                                                                                                          \
    err = hipGetLastError();                                                                                                   \
    return true;                                                                                                               \
  }
  BARTRT_EXT_C_LIST(BARTRT_SLANT_EXT)
#undef BARTRT_SLANT_EXT
  return false;
}

}  // namespace bartrt
