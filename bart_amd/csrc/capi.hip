// extern "C" boundary of libbartrt.so (include/bartrt.h).
#include <cstring>
#include <stdexcept>
#include <string>

#include "../../include/bartrt.h"
#include "engine.hpp"
#include "lbl.hpp"
#include "share.hpp"
#include "step.hpp"
#include "rtc.hpp"
#include "svc.hpp"

#include <climits>
#include <cstdio>
#include <cstdlib>

#include <sys/stat.h>
#include <unistd.h>

using namespace bartrt;

// One of two shapes per process: an engine of its own (g_eng), or -- `shareOpacity` as the chain service,
// svc.hpp -- a client slot (g_cli) of the engine some worker process of the run owns (g_svc set in that one).
static Engine *g_eng = nullptr;
static ChainService *g_svc = nullptr;
static svc::Client *g_cli = nullptr;
static thread_local std::string g_err;

static int fail(int code, const std::string &m) {
  g_err = m;
  return code;
}

template <class F>
static int guarded(F &&f) {
  try {
    return f();
  } catch (const IoError &e) {
    return fail(BARTRT_EIO, e.msg);
  } catch (const HipError &e) {
    return fail(BARTRT_ENODEV, std::string(e.what) + ": " + hipGetErrorString(e.e));
  } catch (const svc::Error &e) {
    return fail(e.code, e.msg);
  } catch (const std::exception &e) {
    return fail(BARTRT_EINVAL, e.what());
  }
}

#define NEED_ENGINE()                                                                                              \
  if (!g_eng)                                                                                                      \
  return g_cli ? fail(BARTRT_ENOTSUP, std::string(__func__) +                                                      \
                                          ": this process is a client of the shareOpacity chain service (the engine lives in " \
                                          "process " + std::to_string(g_cli->seg.hdr()->owner_pid.load()) +         \
                                          "); the reference module's eight calls and the plain getters are served -- " \
                                          "BARTRT_SHARE_MODE=ipc (or off) gives every process its own engine")      \
               : fail(BARTRT_EINVAL, "engine not initialised: call bartrt_init first")
// the calls a service client answers itself
#define CLIENT_OR_ENGINE() \
  if (!g_eng && !g_cli) return fail(BARTRT_EINVAL, "engine not initialised: call bartrt_init first")

// ---- start-up: which shape this process takes ---------------------------
namespace {

struct InitArgs {
  std::string cfile;
  int shard_rank = 0, shard_n = 1, device = -1;
  bool no_service = false;
};

InitArgs parse_init_args(int argc, const char **argv) {
  InitArgs a;
  for (int i = 1; i < argc; i++) {
    const std::string s = argv[i];
    if ((s == "-c" || s == "--config_file") && i + 1 < argc) a.cfile = argv[++i];
    else if (s == "--shard" && i + 2 < argc) { a.shard_rank = std::atoi(argv[++i]); a.shard_n = std::atoi(argv[++i]); }
    else if (s == "--device" && i + 1 < argc) a.device = std::atoi(argv[++i]);
    else if (s == "--no-service") a.no_service = true;
  }
  return a;
}

// How many GPUs this process can see, WITHOUT a HIP call (a service client never makes one): the entries of
// HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES if set, else the KFD topology nodes that have SIMDs.  0: unknown.
int visible_gpu_count() {
  for (const char *name : {"HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"}) {
    const char *v = std::getenv(name);
    if (v && *v) {
      int n = 1;
      for (const char *c = v; *c; c++) n += *c == ',';
      return n;
    }
  }
  int n = 0;
  for (int node = 0; node < 256; node++) {
    char path[128];
    std::snprintf(path, sizeof path, "/sys/class/kfd/kfd/topology/nodes/%d/properties", node);
    FILE *f = std::fopen(path, "r");
    if (!f) break;
    char key[64];
    long long val;
    while (std::fscanf(f, "%63s %lld", key, &val) == 2)
      if (std::strcmp(key, "simd_count") == 0) { n += val > 0; break; }
    std::fclose(f);
  }
  return n;
}

std::string service_key(const InitArgs &a) {
  struct stat st;
  if (stat(a.cfile.c_str(), &st) != 0) throw IoError{"cannot open configuration file '" + a.cfile + "'"};
  char rp[PATH_MAX];
  const std::string real = realpath(a.cfile.c_str(), rp) ? std::string(rp) : a.cfile;
  // (no HIP call here: a client never makes one -- the GPU is named by what selects it, resolved the way
  // Engine::setup resolves it: LOCAL_RANK modulo the visible devices, so that ten workers with ten LOCAL_RANKs on a
  // one-GPU node meet on one engine instead of electing themselves ten times)
  const char *lr = std::getenv("LOCAL_RANK"), *hv = std::getenv("HIP_VISIBLE_DEVICES"), *rv = std::getenv("ROCR_VISIBLE_DEVICES");
  int dev = a.device;
  if (dev < 0) {
    dev = lr ? std::atoi(lr) : 0;
    const int ndev = visible_gpu_count();
    if (ndev > 0) dev %= ndev;
  }
  // (the library's own id: workers on different builds of libbartrt do not share an engine)
  return real + "|" + std::to_string((long long)st.st_size) + "|" + std::to_string((long long)st.st_mtime) + "|" +
         std::to_string((long long)getuid()) + "|dev " + std::to_string(dev) + "|hip " + (hv ? hv : "") + "|rocr " + (rv ? rv : "") +
         "|shard " + std::to_string(a.shard_rank) + "/" + std::to_string(a.shard_n) + "|svc v3|" + bartrt_build_id();
}

std::string what_failed() {
  try {
    throw;
  } catch (const IoError &e) { return e.msg;
  } catch (const HipError &e) { return std::string(e.what) + ": " + hipGetErrorString(e.e);
  } catch (const svc::Error &e) { return e.msg;
  } catch (const std::exception &e) { return e.what();
  } catch (...) { return "unknown error"; }
}

void teardown(double wait_s) {
  if (g_cli) {
    g_cli->detach();
    g_cli->seg.unmap();
    delete g_cli;
    g_cli = nullptr;
  }
  if (g_svc) {
    ChainService *s = g_svc;
    g_svc = nullptr;
    s->shutdown(wait_s);
  }
  if (g_eng) {
    (void)hipDeviceSynchronize();
    delete g_eng;
    g_eng = nullptr;
  }
}

// a process that exits without trm.free_memory(): the dispatcher thread must not outlive the HIP runtime
void at_exit() {
  try { teardown(0.0); } catch (...) {}
}

void start_service(int argc, const char **argv, const InitArgs &ia) {
  const std::string name = svc::hashed_name("bartrt_svc_", service_key(ia));
  svc::Segment seg;
  const bool owner = svc::elect(name, seg);
  auto *cli = new svc::Client;
  try {
    if (owner) {
      Engine *e = new Engine();
      e->share_mode = kShareOff;      // (the grid is this process's own: nobody maps it)
      try {
        e->init(argc, argv);
        g_svc = ChainService::start(std::move(seg), e);
      } catch (...) {
        const std::string why = what_failed();
        svc::retire(seg, why.c_str());
        seg.unmap();
        delete e;
        throw;
      }
      static bool hooked = false;
      if (!hooked) { std::atexit(at_exit); hooked = true; }
      // this process's own calls go through a slot like everybody's (a second mapping of the segment)
      cli->seg.name = name;
      cli->seg.fd = shm_open(name.c_str(), O_RDWR | O_CLOEXEC, 0600);
      if (cli->seg.fd < 0) throw svc::Error{BARTRT_EIO, "shareOpacity: cannot reopen " + name};
      cli->seg.remap(g_svc->seg.hdr()->total_bytes);
      cli->attach(false);
      svc::open_for_clients(g_svc->seg);
    } else {
      cli->seg = seg;
      cli->attach(true);
    }
  } catch (...) {
    cli->detach();
    cli->seg.unmap();
    delete cli;
    if (g_svc) { ChainService *s = g_svc; g_svc = nullptr; s->shutdown(0.0); }
    throw;
  }
  g_cli = cli;
}

}  // namespace

extern "C" {

const char *bartrt_last_error(void) { return g_err.c_str(); }

int bartrt_init(int argc, const char **argv) {
  if (argc < 1 || !argv) return fail(BARTRT_EINVAL, "bartrt_init: empty argv");
  return guarded([&] {
    teardown(svc::env_num("BARTRT_SHARE_WAIT_S", 60.0));
    const InitArgs ia = parse_init_args(argc, argv);
    if (ia.cfile.empty()) throw IoError{"transit_init: no '-c <configuration file>' in argv"};
    const int mode = resolve_share_mode(read_tcfg(ia.cfile), ia.no_service);
    if (mode == kShareService) {
      start_service(argc, argv, ia);
      return BARTRT_OK;
    }
    Engine *e = new Engine();
    e->share_mode = mode;
    try {
      e->init(argc, argv);
    } catch (...) {
      delete e;
      throw;
    }
    g_eng = e;
    return BARTRT_OK;
  });
}

int bartrt_free_memory(void) {
  return guarded([&] {
    // (the owner of a chain service keeps serving until the other workers have let go, BARTRT_SHARE_WAIT_S at most)
    teardown(svc::env_num("BARTRT_SHARE_WAIT_S", 60.0));
    return BARTRT_OK;
  });
}

int bartrt_get_no_samples(void) {
  CLIENT_OR_ENGINE();
  return g_cli ? g_cli->info.Wfull : g_eng->Wfull;
}

int bartrt_get_waveno_arr(double *out, int n) {
  CLIENT_OR_ENGINE();
  const std::vector<double> &wn = g_cli ? g_cli->info.wn_full : g_eng->wn_full;
  if (!out || n != (int)wn.size()) return fail(BARTRT_EINVAL, "get_waveno_arr: n must equal get_no_samples()");
  std::memcpy(out, wn.data(), sizeof(double) * n);
  return BARTRT_OK;
}

// (a service client keeps its setters to itself: they ride with each of ITS profiles, per walker of the batch)
int bartrt_set_radius(double r_km) {
  CLIENT_OR_ENGINE();
  if (!(r_km > 0)) return fail(BARTRT_EINVAL, "set_radius: radius must be positive");
  if (g_cli) g_cli->over[0] = r_km * 1e5;
  else g_eng->refradius = r_km * 1e5;
  return BARTRT_OK;
}

int bartrt_set_cloudtop(double logp) {
  CLIENT_OR_ENGINE();
  if (g_cli) { g_cli->over[1] = std::pow(10.0, logp) * 1e6; return BARTRT_OK; }
  g_eng->has_cloud = 1;
  g_eng->cloudtop = std::pow(10.0, logp) * 1e6;
  return BARTRT_OK;
}

int bartrt_set_scattering(int flag, double value) {
  CLIENT_OR_ENGINE();
  if (flag < 0 || flag > 2) return fail(BARTRT_EINVAL, "set_scattering: flag must be 0, 1 or 2");
  if (g_cli) { g_cli->scat_flag = flag; g_cli->over[2] = value; return BARTRT_OK; }
  g_eng->scat_flag = flag;
  g_eng->scat_value = value;
  return BARTRT_OK;
}

int bartrt_set_integ(int rule) {
  NEED_ENGINE();
  if (rule < 0 || rule > 2) return fail(BARTRT_EINVAL, "set_integ: rule must be 0 (transmittance), 1 (simpson) or 2 (trapz_tau)");
  g_eng->integ = rule;
  return BARTRT_OK;
}

int bartrt_set_cut(int slant) {
  NEED_ENGINE();
  if (slant != 0 && slant != 1) return fail(BARTRT_EINVAL, "set_cut: 0 (vertical) or 1 (slant)");
  g_eng->cut_slant = slant != 0;
  return BARTRT_OK;
}

int bartrt_set_kernel_by(int local) {
  NEED_ENGINE();
  if (local != 0 && local != 1) return fail(BARTRT_EINVAL, "set_kernel_by: 0 (whole grid) or 1 (local block)");
  g_eng->kernel_by_local = local != 0;
  return BARTRT_OK;
}

int bartrt_get_kernel_by(int *local) {
  NEED_ENGINE();
  if (!local) return fail(BARTRT_EINVAL, "get_kernel_by: null output pointer");
  *local = g_eng->kernel_by_local ? 1 : 0;
  return BARTRT_OK;
}

int bartrt_get_cut(int *slant) {
  CLIENT_OR_ENGINE();
  if (!slant) return fail(BARTRT_EINVAL, "get_cut: null output pointer");
  *slant = g_cli ? g_cli->info.cut_slant : (g_eng->cut_slant ? 1 : 0);
  return BARTRT_OK;
}

int bartrt_get_cia_interp(int *spline) {
  CLIENT_OR_ENGINE();
  if (!spline) return fail(BARTRT_EINVAL, "get_cia_interp: null output pointer");
  *spline = g_cli ? g_cli->info.cia_spline : (g_eng->cia_spline ? 1 : 0);
  return BARTRT_OK;
}

int bartrt_get_share(int *shared, int *owner) {
  CLIENT_OR_ENGINE();
  if (g_cli) {
    if (shared) *shared = 1;
    if (owner) *owner = g_svc ? 1 : 0;
    return BARTRT_OK;
  }
  if (shared) *shared = g_eng->kappa_share ? 1 : 0;
  if (owner) *owner = (g_eng->kappa_share && g_eng->kappa_share->owner) ? 1 : 0;
  return BARTRT_OK;
}

int bartrt_prefetch_profiles_dev(const double *d_prof_next, int nwalkers) {
  NEED_ENGINE();
  if (nwalkers < 0 || (nwalkers > 0 && !d_prof_next)) return fail(BARTRT_EINVAL, "prefetch_profiles_dev: null buffer");
  g_eng->pf_req_prof = nwalkers > 0 ? d_prof_next : nullptr;
  g_eng->pf_req_n = nwalkers;
  return BARTRT_OK;
}

int bartrt_get_integ(int *rule) {
  CLIENT_OR_ENGINE();
  if (!rule) return fail(BARTRT_EINVAL, "get_integ: null output pointer");
  *rule = g_cli ? g_cli->info.integ : g_eng->integ;
  return BARTRT_OK;
}

int bartrt_get_service(int *mode, int *owner_pid, int *slot, int *nclients) {
  CLIENT_OR_ENGINE();
  if (g_cli) {
    const svc::Header *h = g_cli->seg.hdr();
    if (mode) *mode = g_svc ? 2 : 1;
    if (owner_pid) *owner_pid = h->owner_pid.load();
    if (slot) *slot = g_cli->slot;
    if (nclients) {
      int n = 0;
      for (int i = 0; i < h->maxclients; i++) n += g_cli->seg.slot(i)->pid.load() != 0;
      *nclients = n;
    }
  } else {
    if (mode) *mode = 0;
    if (owner_pid) *owner_pid = (int)getpid();
    if (slot) *slot = -1;
    if (nclients) *nclients = 0;
  }
  return BARTRT_OK;
}

int bartrt_get_service_stats(unsigned long long *nlaunches, unsigned long long *nprofiles, unsigned long long *nfull) {
  if (!g_cli) return fail(BARTRT_EINVAL, "get_service_stats: this process is not on a chain service");
  const svc::Header *h = g_cli->seg.hdr();
  if (nlaunches) *nlaunches = h->nbatches.load();
  if (nprofiles) *nprofiles = h->nserved.load();
  if (nfull) *nfull = h->nfull.load();
  return BARTRT_OK;
}

int bartrt_get_service_gathered(unsigned long long *ngathered) {
  if (!g_cli) return fail(BARTRT_EINVAL, "get_service_gathered: this process is not on a chain service");
  if (ngathered) *ngathered = g_cli->seg.hdr()->ngathered.load();
  return BARTRT_OK;
}

int bartrt_get_nlayers(void) { CLIENT_OR_ENGINE(); return g_cli ? g_cli->info.L : g_eng->L; }
int bartrt_get_nspecies(void) { CLIENT_OR_ENGINE(); return g_cli ? g_cli->info.S : g_eng->S; }
int bartrt_get_nprof(void) { CLIENT_OR_ENGINE(); return g_cli ? g_cli->info.nprof() : (g_eng->S + 1) * g_eng->L; }

int bartrt_get_local_range(int *lo, int *hi) {
  CLIENT_OR_ENGINE();
  if (lo) *lo = g_cli ? g_cli->info.lo : g_eng->lo;
  if (hi) *hi = g_cli ? g_cli->info.hi : g_eng->hi;
  return BARTRT_OK;
}

int bartrt_get_species(char *buf, int buflen) {
  CLIENT_OR_ENGINE();
  std::string s;
  if (g_cli) s = g_cli->info.species;
  else for (auto &n : g_eng->atm.species) s += (s.empty() ? "" : " ") + n;
  if (!buf || (int)s.size() + 1 > buflen) return fail(BARTRT_EINVAL, "get_species: buffer too small");
  std::memcpy(buf, s.c_str(), s.size() + 1);
  return BARTRT_OK;
}

int bartrt_get_pressure(double *out, int n) {
  CLIENT_OR_ENGINE();
  const std::vector<double> &press = g_cli ? g_cli->info.press : g_eng->atm.press;
  if (!out || n != (int)press.size()) return fail(BARTRT_EINVAL, "get_pressure: n must equal the layer count");
  std::memcpy(out, press.data(), sizeof(double) * n);
  return BARTRT_OK;
}

// host-buffer calls up to this size skip the staging copies (see below)
static constexpr size_t kZeroCopyBytes = 512 * 1024;

// trm.run_transit of a chain-service client: the profile goes into this process's slot, the dispatcher of the
// owning process launches it together with the other workers' (svc_core.hpp); a batch from ONE client is posted
// profile by profile (the batched callers own their engine: bart_amd.engine initialises with --no-service)
static int client_run(const double *prof, int nwalkers, int nprof, double *spec, int nwave, unsigned char *ok) {
  svc::Client *c = g_cli;
  if (!prof || !spec || nwalkers < 0) return fail(BARTRT_EINVAL, "run_transit: null buffer");
  if (nprof != c->info.nprof()) return fail(BARTRT_EINVAL, "run_transit: profile length must be (nspecies+1)*nlayers");
  const int Wl = c->info.Wl();
  if (nwave != Wl && nwave != c->info.Wfull)
    return fail(BARTRT_EINVAL, "run_transit: nwave must equal get_no_samples() (or the shard size)");
  return guarded([&] {
    const size_t off = nwave == Wl ? 0 : (size_t)c->info.lo;
    for (int w = 0; w < nwalkers; w++)
      c->call(prof + (size_t)w * nprof, spec + (size_t)w * nwave + off, ok ? ok + w : nullptr);
    return BARTRT_OK;
  });
}

int bartrt_run_transit_batch(const double *prof, int nwalkers, int nprof,
                             double *spec, int nwave, unsigned char *ok) {
  if (g_cli) return client_run(prof, nwalkers, nprof, spec, nwave, ok);
  NEED_ENGINE();
  Engine *e = g_eng;
  if (!prof || !spec || nwalkers < 0) return fail(BARTRT_EINVAL, "run_transit: null buffer");
  if (nprof != (e->S + 1) * e->L)
    return fail(BARTRT_EINVAL, "run_transit: profile length must be (nspecies+1)*nlayers");
  const int Wl = e->W();
  if (nwave != Wl && nwave != e->Wfull)
    return fail(BARTRT_EINVAL, "run_transit: nwave must equal get_no_samples() (or the shard size)");
  if (nwalkers == 0) return BARTRT_OK;
  return guarded([&] {
    e->ensure_walkers(nwalkers);
    const size_t pb = sizeof(double) * (size_t)nwalkers * nprof;
    const size_t sb = sizeof(double) * (size_t)nwalkers * Wl;
    e->ensure_pin(pb + sb + nwalkers);
    std::memcpy(e->h_pin, prof, pb);
    double *hs = e->h_pin + (size_t)nwalkers * nprof;
    unsigned char *hok = reinterpret_cast<unsigned char *>(hs + (size_t)nwalkers * Wl);
    if (pb + sb <= kZeroCopyBytes) {
      // a walker or a few: the kernels read the profiles from, and write the
      // spectra to, the pinned host buffer themselves (three copy operations
      // cost more than the kernels at this size)
      void *dev = nullptr;
      HIPCHK(hipHostGetDevicePointer(&dev, e->h_pin, 0));
      double *dp = static_cast<double *>(dev);
      e->run_dev(dp, nwalkers, dp + (size_t)nwalkers * nprof,
                 reinterpret_cast<unsigned char *>(dp + (size_t)nwalkers * nprof + (size_t)nwalkers * Wl),
                 e->stream, false);
      e->last_prof = dp;   // stays valid until the next host-buffer call
      e->last_n = nwalkers;
    } else {
      e->last_prof = e->d_prof;
      e->last_n = nwalkers;
      HIPCHK(hipMemcpyAsync(e->d_prof, e->h_pin, pb, hipMemcpyHostToDevice, e->stream));
      e->run_dev(e->d_prof, nwalkers, e->d_spec, e->d_ok, e->stream, false);
      HIPCHK(hipMemcpyAsync(hs, e->d_spec, sb, hipMemcpyDeviceToHost, e->stream));
      HIPCHK(hipMemcpyAsync(hok, e->d_ok, nwalkers, hipMemcpyDeviceToHost, e->stream));
    }
    e->wait(e->stream);
    const size_t off = nwave == Wl ? 0 : (size_t)e->lo;
    for (int w = 0; w < nwalkers; w++)
      std::memcpy(spec + (size_t)w * nwave + off, hs + (size_t)w * Wl, sizeof(double) * Wl);
    if (ok) {
      std::memcpy(ok, hok, nwalkers);
    } else {
      // no flag array to report through (the reference-shaped single call): refuse loudly
      for (int w = 0; w < nwalkers; w++)
        if (!hok[w]) throw std::invalid_argument("run_transit: the profile holds a non-finite or non-positive temperature");
    }
    return BARTRT_OK;
  });
}

int bartrt_run_transit(const double *prof, int nprof, double *spec, int nwave) {
  return bartrt_run_transit_batch(prof, 1, nprof, spec, nwave, nullptr);
}

int bartrt_run_transit_batch_dev(const double *d_prof, int nwalkers, double *d_spec,
                                 unsigned char *d_ok, void *stream) {
  NEED_ENGINE();
  if (!d_prof || !d_spec || nwalkers < 0) return fail(BARTRT_EINVAL, "run_transit_batch_dev: null buffer");
  return guarded([&] {
    hipStream_t st = stream ? (hipStream_t)stream : g_eng->stream;
    // device-buffer calls keep no profile: the optical-depth / intensity getters must not fall
    // back on an older host-buffer call's
    g_eng->last_prof = nullptr;
    g_eng->last_n = 0;
    g_eng->run_dev(d_prof, nwalkers, d_spec, d_ok, st, false);
    return BARTRT_OK;
  });
}

// the profile the optical-depth / intensity getters re-run: walker w of the latest host-buffer call
static const double *latest_profile(Engine *e, int walker, const char *who) {
  if (!e->last_prof)
    throw IoError{std::string(who) + ": no host-buffer spectrum has been computed since the engine's buffers were last "
                  "(re)built (device-buffer calls keep no profile: call bartrt_run_transit with the one in question)"};
  if (walker < 0 || walker >= e->last_n)
    throw IoError{std::string(who) + ": walker index outside the latest call's batch"};
  return e->last_prof + (size_t)walker * (e->S + 1) * e->L;
}

int bartrt_get_tau_of(int walker, double *tau, int *last, int nwave, int nlayers) {
  NEED_ENGINE();
  Engine *e = g_eng;
  if (!tau || nwave != e->W() || nlayers != e->L)
    return fail(BARTRT_EINVAL, "get_tau: shape must be [local samples][nlayers]");
  return guarded([&] {
    // re-run that profile of the latest host-buffer call with the optical-depth output enabled
    e->run_dev(latest_profile(e, walker, "get_tau"), 1, e->d_spec, e->d_ok, e->stream, true);
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(tau, e->d_tau, sizeof(double) * (size_t)nwave * nlayers, hipMemcpyDeviceToHost));
    if (last) HIPCHK(hipMemcpy(last, e->d_last, sizeof(int) * (size_t)nwave, hipMemcpyDeviceToHost));
    return BARTRT_OK;
  });
}

int bartrt_get_atm_profile(double *prof, int nprof) {
  CLIENT_OR_ENGINE();
  if (g_cli) {
    if (!prof || nprof != g_cli->info.nprof()) return fail(BARTRT_EINVAL, "get_atm_profile: bad length");
    std::memcpy(prof, g_cli->info.atm_prof.data(), sizeof(double) * nprof);
    return BARTRT_OK;
  }
  Engine *e = g_eng;
  if (!prof || nprof != (e->S + 1) * e->L) return fail(BARTRT_EINVAL, "get_atm_profile: bad length");
  for (int l = 0; l < e->L; l++) {
    prof[l] = e->atm.temp[l];
    for (int s = 0; s < e->S; s++) prof[(size_t)(s + 1) * e->L + l] = e->atm.abund[(size_t)l * e->S + s];
  }
  return BARTRT_OK;
}

int bartrt_get_radius(double *rad, int nlayers) {
  NEED_ENGINE();
  if (!rad || nlayers != g_eng->L) return fail(BARTRT_EINVAL, "get_radius: bad length");
  return guarded([&] {
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(rad, g_eng->d_rad, sizeof(double) * nlayers, hipMemcpyDeviceToHost));
    return BARTRT_OK;
  });
}

int bartrt_get_nangles(void) { CLIENT_OR_ENGINE(); return g_cli ? g_cli->info.A : g_eng->A; }

int bartrt_get_tau(double *tau, int *last, int nwave, int nlayers) {
  NEED_ENGINE();
  if (g_eng->last_prof && g_eng->last_n > 1)
    return fail(BARTRT_EINVAL, "get_tau: the latest call was a batch; name the walker with bartrt_get_tau_of()");
  return bartrt_get_tau_of(0, tau, last, nwave, nlayers);
}

int bartrt_get_angles(double *deg, int n) {
  CLIENT_OR_ENGINE();
  const std::vector<double> &ang = g_cli ? g_cli->info.angles : g_eng->angles;
  if (!deg || n != (int)ang.size()) return fail(BARTRT_EINVAL, "get_angles: bad length");
  std::memcpy(deg, ang.data(), sizeof(double) * n);
  return BARTRT_OK;
}

int bartrt_get_intensity(double *intens, int nangles, int nwave) {
  NEED_ENGINE();
  if (g_eng->last_prof && g_eng->last_n > 1)
    return fail(BARTRT_EINVAL, "get_intensity: the latest call was a batch; name the walker with bartrt_get_intensity_of()");
  return bartrt_get_intensity_of(0, intens, nangles, nwave);
}

int bartrt_get_intensity_of(int walker, double *intens, int nangles, int nwave) {
  NEED_ENGINE();
  Engine *e = g_eng;
  if (e->solution != 0) return fail(BARTRT_EINVAL, "get_intensity: eclipse geometry only");
  if (!intens || nangles != e->A || nwave != e->W()) return fail(BARTRT_EINVAL, "get_intensity: bad shape");
  return guarded([&] {
    const double *prof = latest_profile(e, walker, "get_intensity");
    e->want_intens = true;
    try {
      e->run_dev(prof, 1, e->d_spec, e->d_ok, e->stream, false);
    } catch (...) { e->want_intens = false; throw; }
    e->want_intens = false;
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(intens, e->d_intens, sizeof(double) * (size_t)nangles * nwave, hipMemcpyDeviceToHost));
    return BARTRT_OK;
  });
}

int bartrt_get_lbl_extinction(const double *prof, int nprof, double *ext, int nlayers, int nwave) {
  NEED_ENGINE();
  Engine *e = g_eng;
  if (!e->lbl) return fail(BARTRT_EINVAL, "get_lbl_extinction: the engine was not set up with a line list");
  if (!prof || !ext || nprof != (e->S + 1) * e->L || nlayers != e->L || nwave != e->W())
    return fail(BARTRT_EINVAL, "get_lbl_extinction: bad shapes");
  return guarded([&] {
    e->ensure_walkers(1);
    HIPCHK(hipMemcpy(e->d_prof, prof, sizeof(double) * nprof, hipMemcpyHostToDevice));
    lbl_extinction(*e, e->d_prof, 1, e->stream);
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(ext, e->lbl->d_ext, sizeof(double) * (size_t)nlayers * nwave, hipMemcpyDeviceToHost));
    return BARTRT_OK;
  });
}

int bartrt_voigt(const double *x, const double *y, double *k, long n) {
  if (n < 0 || (n > 0 && (!x || !y || !k))) return fail(BARTRT_EINVAL, "bartrt_voigt: bad arguments");
  return guarded([&] {
    lbl_voigt_probe(x, y, k, n);
    return BARTRT_OK;
  });
}

int bartrt_timing_begin(void) { return bartrt_timing_begin_sampled(1); }

int bartrt_timing_begin_sampled(int stride) {
  NEED_ENGINE();
  if (stride < 1) return fail(BARTRT_EINVAL, "timing_begin_sampled: stride must be >= 1");
  return guarded([&] {
    // the events of the sampled launches are created here, outside the caller's timed region (created on
    // demand inside it they cost the first window of bench.py 10 us per step: 81 against 71)
    while (g_eng->ev.size() < 4096) {
      hipEvent_t e;
      HIPCHK(hipEventCreate(&e));
      g_eng->ev.push_back(e);
    }
    g_eng->timing = true;
    g_eng->timing_stride = stride;
    g_eng->timing_seen = 0;
    g_eng->ev_used = 0;
    return BARTRT_OK;
  });
}

int bartrt_timing_end(double *kernel_ms, int *nlaunch) {
  NEED_ENGINE();
  return guarded([&] {
    Engine *e = g_eng;
    double tot = 0.0;
    HIPCHK(hipDeviceSynchronize());
    for (int i = 0; i + 1 < e->ev_used; i += 2) {
      float ms = 0.f;
      HIPCHK(hipEventElapsedTime(&ms, e->ev[i], e->ev[i + 1]));
      tot += ms;
    }
    if (kernel_ms) *kernel_ms = tot;
    if (nlaunch) *nlaunch = e->ev_used / 2;
    e->timing = false;
    e->ev_used = 0;
    return BARTRT_OK;
  });
}

int bartrt_walked_begin(void) {
  NEED_ENGINE();
  g_eng->want_walked = true;
  g_eng->walked_nwalkers = 0;
  return BARTRT_OK;
}

int bartrt_walked_end(int *walked, int cap, int *nwalkers, int *ncolumns, int *wn_per_column,
                      char *kernel, int kernel_len) {
  NEED_ENGINE();
  return guarded([&] {
    Engine *e = g_eng;
    e->want_walked = false;
    HIPCHK(hipDeviceSynchronize());
    const int n = e->walked_nwalkers, nc = e->walked_info.ncolumns;
    if (nwalkers) *nwalkers = n;
    if (ncolumns) *ncolumns = nc;
    if (wn_per_column) *wn_per_column = e->walked_info.wn_per_column;
    if (kernel && kernel_len > 0) {
      std::snprintf(kernel, (size_t)kernel_len, "%s%s%s", e->walked_info.kernel, e->walked_info.rtc ? " [instantiated at run time]" : "",
                    e->walked_info.prep_folded ? " [prepares its own walkers]" : "");
    }
    if (walked && n > 0) {
      if ((size_t)cap < (size_t)n * nc) throw std::invalid_argument("walked_end: buffer too small");
      HIPCHK(hipMemcpy(walked, e->d_walked, sizeof(int) * (size_t)n * nc, hipMemcpyDeviceToHost));
    }
    return BARTRT_OK;
  });
}

int bartrt_get_rtc_stats(int *available, int *compiled, int *from_disk, int *failed, double *compile_seconds) {
  const RtcStats st = rtc_stats();
  if (available) *available = rtc_available() ? 1 : 0;
  if (compiled) *compiled = st.compiled;
  if (from_disk) *from_disk = st.from_disk;
  if (failed) *failed = st.failed;
  if (compile_seconds) *compile_seconds = st.compile_seconds;
  return BARTRT_OK;
}

int bartrt_rtc_compile(const char *expr, int ilp, long *code_bytes) {
  if (!expr) return fail(BARTRT_EINVAL, "rtc_compile: null expression");
  std::string why;
  const long n = rtc_compile_only(expr, ilp != 0, why);
  if (code_bytes) *code_bytes = n;
  return n < 0 ? fail(BARTRT_ENOTSUP, why) : BARTRT_OK;
}

double bartrt_algorithmic_bytes(int nwalkers) {
  if (!g_eng) return 0.0;
  Engine *e = g_eng;
  const double L = e->L, W = e->W(), M = e->M, C = e->C, S = e->S;
  return nwalkers * (2.0 * L * W * M * 8.0 + 2.0 * L * W * C * 8.0 + (S + 1) * L * 8.0 + W * 8.0);
}

// ---- per-step converters (step.hip) ------------------------------------
int bartrt_step_setup(const double *ptargs5, int tint_thorngren, int pttype,
                      double tmin, double tmax,
                      const double *abund, int nmolfit, const int *imol,
                      int nfilters, const int *idx0, const int *npts,
                      const double *nifilter, const double *istarfl,
                      double rprs, int solution) {
  NEED_ENGINE();
  return guarded([&] {
    step_setup(*g_eng, ptargs5, tint_thorngren, pttype, tmin, tmax, abund, nmolfit, imol,
               nfilters, idx0, npts, nifilter, istarfl, rprs, solution);
    return BARTRT_OK;
  });
}

int bartrt_step_set_ebalance(int on, double e_in, double e_fac) {
  NEED_ENGINE();
  return guarded([&] {
    step_set_ebalance(*g_eng, on, e_in, e_fac);
    return BARTRT_OK;
  });
}

int bartrt_step_profiles_dev(const double *d_params, int nwalkers, int npars,
                             double *d_prof, int *d_status, void *stream) {
  NEED_ENGINE();
  if (!g_eng->step) return fail(BARTRT_EINVAL, "step_profiles: call bartrt_step_setup first");
  if (!d_params || !d_prof || !d_status) return fail(BARTRT_EINVAL, "step_profiles: null buffer");
  return guarded([&] {
    hipStream_t st = stream ? (hipStream_t)stream : g_eng->stream;
    step_profiles_dev(*g_eng, d_params, nwalkers, npars, d_prof, d_status, st);
    return BARTRT_OK;
  });
}

int bartrt_step_bandflux_dev(const double *d_spec_full, int nwalkers, int *d_status,
                             double *d_bandflux, void *stream) {
  NEED_ENGINE();
  if (!g_eng->step) return fail(BARTRT_EINVAL, "step_bandflux: call bartrt_step_setup first");
  if (!d_spec_full || !d_bandflux || !d_status) return fail(BARTRT_EINVAL, "step_bandflux: null buffer");
  return guarded([&] {
    hipStream_t st = stream ? (hipStream_t)stream : g_eng->stream;
    step_bandflux_dev(*g_eng, d_spec_full, nwalkers, d_status, d_bandflux, st);
    return BARTRT_OK;
  });
}

int bartrt_step_batch_dev(const double *d_params, int nwalkers, int npars,
                          double *d_bandflux, int *d_status, double *d_spec,
                          void *stream) {
  NEED_ENGINE();
  if (!g_eng->step) return fail(BARTRT_EINVAL, "step_batch: call bartrt_step_setup first");
  return guarded([&] {
    hipStream_t st = stream ? (hipStream_t)stream : g_eng->stream;
    step_run_dev(*g_eng, d_params, nwalkers, npars, d_bandflux, d_status, d_spec, st);
    return BARTRT_OK;
  });
}

int bartrt_step_set_extras(int nrad, int ncloud, int nray) {
  NEED_ENGINE();
  return guarded([&] {
    step_set_extras(*g_eng, nrad, ncloud, nray);
    return BARTRT_OK;
  });
}

int bartrt_step_set_carry(int on) {
  NEED_ENGINE();
  return guarded([&] {
    step_set_carry(*g_eng, on);
    return BARTRT_OK;
  });
}

int bartrt_step_batch(const double *params, int nwalkers, int npars,
                      double *bandflux, int *status) {
  NEED_ENGINE();
  if (!g_eng->step) return fail(BARTRT_EINVAL, "step_batch: call bartrt_step_setup first");
  if (!params || !bandflux || nwalkers < 0) return fail(BARTRT_EINVAL, "step_batch: null buffer");
  return guarded([&] {
    step_run_host(*g_eng, params, nwalkers, npars, bandflux, status);
    return BARTRT_OK;
  });
}

int bartrt_mcmc_run(int nchains, int npars, long nsteps, const double *params, const double *pmin,
                    const double *pmax, const double *stepsize, int ndata, const double *data,
                    const double *uncert, int snooker, unsigned long long seed, double *chain,
                    double *chisq, long *naccept, long *nbad) {
  NEED_ENGINE();
  if (!g_eng->step) return fail(BARTRT_EINVAL, "mcmc_run: call bartrt_step_setup first");
  if (!params || !pmin || !pmax || !stepsize || !data || !uncert || !chain || !chisq)
    return fail(BARTRT_EINVAL, "mcmc_run: null buffer");
  return guarded([&] {
    mcmc_run(*g_eng, nchains, npars, nsteps, params, pmin, pmax, stepsize, ndata, data, uncert,
             snooker, seed, chain, chisq, naccept, nbad);
    return BARTRT_OK;
  });
}

}  // extern "C"
