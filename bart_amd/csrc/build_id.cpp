// bartrt_build_id(): a hash of the CODE the eclipse RT launch is made of (the RT objects' device code and host text,
// bart_amd/build.py code_id), compiled in after everything else.  Profiler figures committed under profiles/ carry it;
// bench.py quotes them only when the loaded library answers the same id.
#include "build_id.inc"
extern "C" const char *bartrt_build_id(void) { return BARTRT_BUILD_ID; }
