// The chain service on the GPU engine (protocol: svc_core.hpp).
#pragma once
#include <string>
#include <thread>

#include "svc_core.hpp"

namespace bartrt {

struct Engine;

// `shareOpacity` (code/makecfg.py:106-107, BART.py:259-262) has three readings here:
//   kShareService (default with the key): one engine per (cfg, GPU, wavenumber block), owned by the first worker
//                 process to initialise; every worker is a client of its dispatcher (svc_core.hpp) -- one HIP
//                 context, one grid, one batched launch per MCMC step;
//   kShareIpc     every worker runs its own engine, the grid is one HBM allocation mapped through HIP IPC (share.hpp);
//   kShareOff     every worker for itself.
// Chosen by `--share-mode service|ipc|off` in transit_init's argv, else BARTRT_SHARE_MODE, else the cfg key
// (BARTRT_SHARE_OPACITY=0/1 overrides the key as before; BARTRT_SERVICE=1 asks for the service without the key).
enum ShareMode { kShareOff = 0, kShareIpc = 1, kShareService = 2 };

// The owner's side: the engine, the registered data area of the segment and the dispatcher thread.
struct ChainService : svc::Backend {
  svc::Segment seg;
  Engine *eng = nullptr;
  svc::Dispatcher disp;
  std::thread th;
  bool registered = false;       // the segment's data area is page-locked and mapped on the GPU
  double *d_prof = nullptr, *d_spec = nullptr, *d_over = nullptr;   // device views of the segment (registered)
  unsigned char *d_ok = nullptr;
  double *d_over_stage = nullptr;   // unregistered fallback
  size_t direct_spec_bytes = 0;     // spectra of a launch up to this size are written to host memory by the kernel itself
  // how the dispatcher learns that a launch has finished: 0 hipStreamSynchronize, 1 a stream write of a sequence
  // number into the (registered) segment that the thread polls, 2 an event it polls (BARTRT_SVC_SYNC)
  int sync_mode = 0;
  uint32_t flag_seq = 0;
  uint32_t *d_flag = nullptr;
  int32_t *d_list = nullptr;        // device view of the segment's slot list (gathered launches)
  void *ev_done = nullptr;
  void wait_done();

  // takes over `seg` (elected, one page) and `e` (initialised); publishes, registers, starts the thread
  static ChainService *start(svc::Segment &&seg, Engine *e);
  // keeps serving until the other clients have detached (or wait_s passed), then stops; deletes the engine
  void shutdown(double wait_s);
  void run(const int *slots, int n, int nominal, int scat_flag, bool any_over, bool any_cloud) override;
  ~ChainService() override {}
};

}  // namespace bartrt
