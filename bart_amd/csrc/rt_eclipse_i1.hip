// Specialised eclipse kernels under integration rule 1 (integ.hpp): the template
// definitions are in rt_eclipse.hpp; one translation unit per rule keeps the
// build parallel.
#include "rt_eclipse.hpp"

namespace bartrt {
template bool launch_rt_spec<1>(const RtArgs &, int, hipStream_t, const std::string &, bool, bool, hipError_t &,
                                RtLaunchInfo *, const PrepArgs *);
}  // namespace bartrt
