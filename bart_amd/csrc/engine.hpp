// Host-side engine state: parsed inputs, HBM-resident tables, workspaces.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "io.hpp"
#include "kernels.hpp"

namespace bartrt {

hipError_t launch_prep(const PrepArgs &a, hipStream_t st);
hipError_t launch_rt(const RtArgs &a, int block, hipStream_t st, RtLaunchInfo *info = nullptr);
hipError_t launch_rt_folded(const RtArgs &a, const PrepArgs &prep, int block, hipStream_t st, RtLaunchInfo *info, bool *folded);
hipError_t launch_transit(const RtArgs &a, hipStream_t st);
hipError_t launch_chord_table(const PrepArgs &a, hipStream_t st);  // transit geometry, after launch_prep
hipError_t launch_grid_transpose(const double *src, double *dst, long planes, int M, int W, hipStream_t st);

int parse_integ(const std::string &v);  // "0" / "transmittance", "1" / "simpson", "2" / "trapz_tau"
// what `shareOpacity` means for this process (ShareMode, svc.hpp): cfg key, BARTRT_SHARE_OPACITY, BARTRT_SHARE_MODE,
// BARTRT_SERVICE; no_service: the caller needs the engine in its own process (the service reading becomes IPC)
int resolve_share_mode(const TCfg &cfg, bool no_service);

struct TableShare;  // the opacity grid shared between processes (share.hpp)
struct StepArgs;  // converters around the engine (step.hip)
struct Lbl;       // line-by-line extinction (lbl.hip)

struct Engine {
  // configuration
  TCfg cfg;
  Atm atm;
  MolInfo mol;
  int L = 0, S = 0, M = 0, Nt = 0, C = 0, A = 0;
  int Wfull = 0, lo = 0, hi = 0;  // this process holds [lo, hi) of the grid
  int W() const { return hi - lo; }
  std::vector<double> wn_full, tgrid, mass, angles;
  std::vector<int> opmol;
  double toomuch = 20.0, gsurf = 0, refpress = 0, refradius = 0;
  int scat_flag = 0, iH2 = -1, iHe = -1, has_cloud = 0;
  int solution = 0;       // 0 eclipse (emergent flux), 1 transit (modulation)
  int integ = 0;          // integration rule of the eclipse geometry (integ.hpp); cfg `integ`, BARTRT_INTEG
  bool cut_slant = false; // cfg `cut slant` / BARTRT_CUT: the toomuch cut per ray angle, on its slant depth (C19)
  // Sharded engines (--shard): is the kernel variant chosen by this block's own columns (true, the default since round 5:
  // a WASP-12b block of 303 samples x 10 walkers takes the layer-parallel kernel, 16 us, instead of the single-wave
  // kernel's latency floor, 52 us; spectra agree with the unsharded run's to 4e-16) or by the WHOLE grid's (false: cfg
  // `kernel_by whole` / BARTRT_KERNEL_BY=whole / bartrt_set_kernel_by -- the blocks then are the unsharded run's bits)
  bool kernel_by_local = true;
  int sel_walkers = 0;    // > 0: the walker count the kernel variant is chosen for instead of the batch's own (RtArgs::nsel)
  bool cia_spline = false; // cfg `cia_interp spline` / BARTRT_CIA_INTERP: natural cubic splines in wavenumber and T (C20)
  double starrad = 0;     // cm, transit geometry
  double scat_value = 0, cloudtop = 0;
  double cloud_rup = 0, cloud_rdown = 0, cloud_ext = 0;  // radius-ramp cloud (cm, cm, cm-1); 0 = none
  bool transparent = false;                              // transit geometry: no opaque core
  int device = 0;
  // device-resident inputs
  double *d_kappa = nullptr, *d_cia = nullptr, *d_wn = nullptr, *d_wn_full = nullptr;
  double *d_press = nullptr, *d_mass = nullptr, *d_diam = nullptr;
  TableShare *kappa_share = nullptr;  // cfg `shareOpacity`: d_kappa is (or is mapped from) another process's allocation
  int share_mode = -1;                // ShareMode (svc.hpp); -1: resolved from the cfg and the environment by setup()
  double *d_prep_consts = nullptr;  // PrepArgs::consts
  PrepArgs prep{};  // static part filled at init
  RtArgs rt{};
  // workspaces (grown on demand, never inside a timed launch sequence twice)
  int cap_walkers = 0;
  double *d_prof = nullptr, *d_coef = nullptr, *d_spec = nullptr;
  const double *last_prof = nullptr;  // profiles of the latest host-buffer call (get_tau, get_intensity)
  int last_n = 0;                     // ... and how many they are
  idx_t *d_idx = nullptr;
  int *d_kstop = nullptr;
  double *d_rtop = nullptr, *d_ds = nullptr;  // transit geometry workspaces
  double *d_rad = nullptr;     // [cap][L] hydrostatic radii of the last run
  double *d_intens = nullptr;  // [A][W] of the last single-walker run with want_intens
  bool want_intens = false;
  bool lbl_eager = true;       // full extinction first (default); BARTRT_LBL=lazy: fused kernel
  unsigned char *d_ok = nullptr;
  double *d_tau = nullptr;  // [W][L] of the last single-walker run
  int *d_last = nullptr;
  double *h_pin = nullptr;  // pinned staging
  size_t h_pin_bytes = 0;
  hipStream_t stream = nullptr;
  // Host calls that wait for their result (bartrt_run_transit, bartrt_step_batch): the stream writes a sequence number
  // into a word of pinned host memory behind the kernels and the calling thread polls it -- 5-7 us sooner per call than
  // hipStreamSynchronize wakes up (measured in the chain service: 121 against 128 us per ten-walker call).
  // BARTRT_SYNC=stream: hipStreamSynchronize.
  unsigned int *h_flag = nullptr, *d_flag = nullptr;
  unsigned int flag_seq = 0;
  bool sync_poll = true;
  void wait(hipStream_t st);
  // when set, run_chunk calls this instead of launch_prep (the per-step path
  // fuses T(p) + abundances into the same launch, step.hip)
  hipError_t (*prep_hook)(const PrepArgs &, hipStream_t, void *) = nullptr;
  void *prep_hook_ctx = nullptr;
  // per-walker radius / cloud / scattering overrides for the NEXT prep launch only
  // (set by step_profiles_dev, consumed by run_chunk); prep_over_cloud: a cloud-top
  // parameter is among them, so the RT kernels add the deck's surface term
  const double *prep_over_once = nullptr;
  bool prep_over_cloud = false;
  // diagnostics of the next RT launches: layers walked per (walker, kernel column)
  bool want_walked = false;
  int *d_walked = nullptr;
  size_t walked_cap = 0;
  int walked_nwalkers = 0;
  RtLaunchInfo walked_info{};
  // timing of RT launches
  bool timing = false;
  int timing_stride = 1;     // events bracket every timing_stride-th launch (a pair of event
  long timing_seen = 0;      // records costs the stream ~5 us: bench.py samples)
  std::vector<hipEvent_t> ev;
  int ev_used = 0;
  // Prefetched preparation (bartrt_prefetch_profiles_dev): the caller names the NEXT batch's
  // profile buffer; the RT launch of the current call prepares that batch's layer records in
  // extra workgroups (RtArgs::nprep) into the second set of record buffers, and the next
  // call -- if it is for that buffer -- starts on its RT kernel directly.
  const double *pf_req_prof = nullptr;   // requested for the call after the next run
  int pf_req_n = 0;
  const double *pf_have_prof = nullptr;  // records of this batch are in buffer set pf_have_buf
  int pf_have_n = 0, pf_have_buf = 0;
  // ... built on this stream under these settings (everything of PrepArgs a setter can change)
  struct PrepSettings {
    double refradius, gsurf, cloudtop, scat_value, cloud_rup, cloud_rdown, cloud_ext;
    int has_cloud, scat_flag;
    bool operator==(const PrepSettings &o) const {
      return refradius == o.refradius && gsurf == o.gsurf && cloudtop == o.cloudtop && scat_value == o.scat_value &&
             cloud_rup == o.cloud_rup && cloud_rdown == o.cloud_rdown && cloud_ext == o.cloud_ext &&
             has_cloud == o.has_cloud && scat_flag == o.scat_flag;
    }
  };
  hipStream_t pf_have_stream = nullptr;
  PrepSettings pf_have_set{};
  double *d_coef2 = nullptr;
  idx_t *d_idx2 = nullptr;
  int *d_kstop2 = nullptr;
  unsigned char *d_ok2 = nullptr;
  int cap2 = 0;                          // walkers the second set holds
  void *d_slog = nullptr;                // `cut slant`: the single-wave kernels' event log (RtArgs::slog)
  size_t slog_cap = 0;
  // per-step converters
  StepArgs *step = nullptr;
  Lbl *lbl = nullptr;

  ~Engine();
  void init(int argc, const char **argv);
  void setup(const TCfg &cfg_in, int shard_rank, int shard_n);
  void ensure_walkers(int n);
  void ensure_pin(size_t bytes);
  // d_prof_in -> d_spec_out ([n][W]); records events when timing
  void run_dev(const double *d_prof_in, int n, double *d_spec_out, unsigned char *d_okp,
               hipStream_t st, bool want_tau);
  void run_chunk(const double *d_prof_in, int n, double *d_spec_out, unsigned char *d_okp,
                 hipStream_t st, bool want_tau, const double *d_ext, bool lbl_fused = false);
};

struct HipError {
  hipError_t e;
  const char *what;
};
#define HIPCHK(x)                                  \
  do {                                             \
    hipError_t _e = (x);                           \
    if (_e != hipSuccess) throw HipError{_e, #x};  \
  } while (0)

}  // namespace bartrt
