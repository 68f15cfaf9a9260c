// The chain service: one process of a retrieval owns the engine, every worker process posts its
// profile into a shared-memory slot and ONE batched launch serves them all.
//
// The reference runs one BARTfunc worker per chain (examples/WASP-12b/BART.cfg:113: ten), each with
// its own transit instance; MC3 releases them together with one Scatter and collects them with one
// Gather (code/BARTfunc.py:312,399), so their trm.run_transit calls (code/BARTfunc.py:363) arrive
// within microseconds of each other.  Ten HIP contexts time-slicing one GPU serve them one by one;
// here the first worker to initialise on a (cfg, GPU, wavenumber block) becomes the OWNER -- it
// builds the engine and starts a dispatcher thread -- and every worker (the owner's own Python
// thread included) is a CLIENT: no HIP call at all, a slot in a POSIX shared-memory segment, a futex.
//
// This header is the protocol only (POSIX + Linux futex, no HIP): the segment layout, the election
// of the owner under a file lock, the client's call and the dispatcher's loop.  What a batch
// computes is the `SvcBackend` the owner hands to the dispatcher (csrc/svc.hip: the engine);
// tests/svc_harness.cpp runs the same protocol with an arithmetic stand-in on CPU-only machines.
#pragma once
#include <atomic>
#include <cerrno>
#include <chrono>
#include <climits>
#include <cmath>
#include <csignal>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <linux/futex.h>
#include <sys/file.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/syscall.h>
#include <unistd.h>

namespace bartrt {
namespace svc {

struct Error {
  int code;          // BARTRT_* (include/bartrt.h)
  std::string msg;
};
constexpr int kEINVAL = -1, kEIO = -2, kENODEV = -3, kENOTSUP = -4;

using clk = std::chrono::steady_clock;
inline double since(clk::time_point t0) { return std::chrono::duration<double>(clk::now() - t0).count(); }
inline double env_num(const char *name, double dflt) {
  const char *e = std::getenv(name);
  return (e && *e) ? std::atof(e) : dflt;
}
inline bool pid_alive(int pid) { return pid > 0 && (kill(pid, 0) == 0 || errno != ESRCH); }
inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#else
  std::this_thread::yield();
#endif
}

// process-shared futex on a 32-bit word of the segment
inline void futex_wait(std::atomic<uint32_t> *w, uint32_t expected, double seconds) {
  timespec ts{(time_t)seconds, (long)((seconds - std::floor(seconds)) * 1e9)};
  (void)syscall(SYS_futex, reinterpret_cast<uint32_t *>(w), FUTEX_WAIT, expected, &ts, nullptr, 0);
}
inline void futex_wake(std::atomic<uint32_t> *w) {
  (void)syscall(SYS_futex, reinterpret_cast<uint32_t *>(w), FUTEX_WAKE, INT_MAX, nullptr, nullptr, 0);
}

inline std::string hashed_name(const char *prefix, const std::string &key) {
  uint64_t h = 1469598103934665603ull;          // FNV-1a
  for (unsigned char c : key) { h ^= c; h *= 1099511628211ull; }
  char buf[80];
  std::snprintf(buf, sizeof buf, "/%s%016llx", prefix, (unsigned long long)h);
  return buf;
}

// Exclusive lock that serialises who creates / replaces a named segment: flock on
// /dev/shm/<name>.lock.  The file may be unlinked by a holder (`remove_file`); a waiter that
// wakes up holding the lock of an unlinked file starts over on the name's current file.
struct NameLock {
  int fd = -1;
  std::string path;
  explicit NameLock(const std::string &shm_name) : path("/dev/shm" + shm_name + ".lock") {
    for (;;) {
      fd = open(path.c_str(), O_CREAT | O_RDWR | O_CLOEXEC, 0600);
      if (fd < 0) throw Error{kEIO, "shareOpacity: cannot open " + path + ": " + std::strerror(errno)};
      if (flock(fd, LOCK_EX) != 0) { close(fd); throw Error{kEIO, "shareOpacity: flock failed on " + path}; }
      struct stat a, b;
      if (fstat(fd, &a) == 0 && stat(path.c_str(), &b) == 0 && a.st_ino == b.st_ino && a.st_dev == b.st_dev) return;
      close(fd);            // the file we locked is no longer the name's: again
    }
  }
  void remove_file() { (void)unlink(path.c_str()); }
  ~NameLock() { if (fd >= 0) { (void)flock(fd, LOCK_UN); close(fd); } }
  NameLock(const NameLock &) = delete;
  NameLock &operator=(const NameLock &) = delete;
};

// ---- segment layout -------------------------------------------------------------------------
constexpr uint32_t kVersion = 4;
constexpr uint32_t kLoading = 0, kReady = 1, kGone = 2;
constexpr uint32_t kIdle = 0, kPosted = 1, kDone = 2, kFailed = 3;
// One slot per worker process = per MCMC chain (examples/demo/BART_eclipse.cfg:90-91: "number of parallel chains", free in
// the reference).  Default 256 (BARTRT_SVC_MAXCLIENTS; 86 kB of segment each at the headline shape), at most 1 024.
constexpr int kMaxClients = 1024;
constexpr int kDefaultClients = 256;
constexpr size_t kPage = 4096;

struct alignas(128) Slot {
  std::atomic<int32_t> pid;         // 0 = free
  std::atomic<uint32_t> st;         // futex word: kIdle / kPosted / kDone / kFailed
  std::atomic<uint32_t> sleeping;   // the client is (about to be) in futex_wait on st
  int32_t scat_flag;                // -1 = the cfg's; else trm.set_scattering's flag (BARTfunc.py:358,360)
  int32_t rc;                       // kFailed: the error code ...
  char err[200];                    // ... and message
};

struct Header {
  std::atomic<uint32_t> state;      // kLoading / kReady / kGone
  uint32_t version;
  std::atomic<int32_t> owner_pid;
  int32_t maxclients;
  uint64_t total_bytes;             // size of the whole segment once state == kReady
  // what a client needs to answer the reference module's getters without an engine
  int32_t L, S, A, Wfull, lo, hi, integ, cut_slant, cia_spline, solution, device;
  int32_t nprof, Wl;                // (S+1)*L, hi-lo
  std::atomic<int32_t> nhi;         // 1 + the highest slot ever taken: the dispatcher scans [0, nhi) (slots are taken lowest first)
  uint64_t off_slots, off_prof, off_spec, off_over, off_ok, off_flag, off_list, off_info;   // byte offsets from the segment's start
  uint64_t off_wn, off_press, off_atmprof, off_angles, off_species, species_len;
  // dispatcher
  alignas(64) std::atomic<uint32_t> bell;      // futex word: bumped by every post
  std::atomic<uint32_t> asleep;                // the dispatcher is (about to be) in futex_wait on bell
  alignas(64) std::atomic<uint64_t> nbatches;  // launches made ...
  std::atomic<uint64_t> nserved;               // ... and profiles served by them
  std::atomic<uint64_t> nfull;                 // launches that held every registered client
  std::atomic<uint64_t> ngathered;             // launches whose slots were not consecutive (one launch all the same)
  char owner_err[256];                         // kGone after a failed start: why
};
static_assert(sizeof(Header) <= kPage, "Header must fit the first page");
static_assert(std::atomic<uint32_t>::is_always_lock_free && std::atomic<uint64_t>::is_always_lock_free, "lock-free atomics");

inline size_t round_up(size_t n, size_t a) { return (n + a - 1) / a * a; }

// what the owner publishes about the engine (copied into the segment's info area)
struct Info {
  int L = 0, S = 0, A = 0, Wfull = 0, lo = 0, hi = 0, integ = 0, cut_slant = 0, cia_spline = 0, solution = 0, device = 0;
  std::vector<double> wn_full, press, atm_prof, angles;
  std::string species;              // space separated
  int nprof() const { return (S + 1) * L; }
  int Wl() const { return hi - lo; }
};

struct Layout {
  size_t off_slots, off_prof, off_spec, off_over, off_ok, off_flag, off_list, off_wn, off_press, off_atmprof, off_angles, off_species, total;
  static Layout of(const Info &i, int maxclients) {
    Layout l{};
    size_t o = kPage;
    l.off_slots = o; o += round_up(sizeof(Slot) * (size_t)maxclients, kPage);
    // the data the backend reads and writes (GPU-visible when the owner registers it): page aligned
    l.off_prof = o; o += round_up(sizeof(double) * (size_t)maxclients * i.nprof(), kPage);
    l.off_spec = o; o += round_up(sizeof(double) * (size_t)maxclients * i.Wl(), kPage);
    l.off_over = o; o += round_up(sizeof(double) * (size_t)maxclients * 3, 256);
    l.off_ok = o; o += round_up((size_t)maxclients, 256);
    l.off_flag = o; o += 256;            // a word the GPU writes when a launch has finished (csrc/svc.hip)
    l.off_list = o; o += round_up(sizeof(int32_t) * (size_t)maxclients, 256);   // the slots of the launch in flight (gathered launches)
    o = round_up(o, kPage);
    l.off_wn = o; o += sizeof(double) * i.wn_full.size();
    l.off_press = o; o += sizeof(double) * i.press.size();
    l.off_atmprof = o; o += sizeof(double) * i.atm_prof.size();
    l.off_angles = o; o += sizeof(double) * i.angles.size();
    l.off_species = o; o += i.species.size() + 1;
    l.total = round_up(o, kPage);
    return l;
  }
};

// ---- a mapped segment -----------------------------------------------------------------------
struct Segment {
  std::string name;
  int fd = -1;
  char *base = nullptr;
  size_t mapped = 0;
  Header *hdr() const { return reinterpret_cast<Header *>(base); }
  Slot *slot(int i) const { return reinterpret_cast<Slot *>(base + hdr()->off_slots) + i; }
  double *prof(int i) const { return reinterpret_cast<double *>(base + hdr()->off_prof) + (size_t)i * hdr()->nprof; }
  double *spec(int i) const { return reinterpret_cast<double *>(base + hdr()->off_spec) + (size_t)i * hdr()->Wl; }
  double *over(int i) const { return reinterpret_cast<double *>(base + hdr()->off_over) + (size_t)i * 3; }
  unsigned char *ok(int i) const { return reinterpret_cast<unsigned char *>(base + hdr()->off_ok) + i; }
  int32_t *list() const { return reinterpret_cast<int32_t *>(base + hdr()->off_list); }
  // the part the backend's kernels touch: [data_begin, data_begin + data_bytes)
  char *data_begin() const { return base + hdr()->off_prof; }
  size_t data_bytes() const { return hdr()->off_wn - hdr()->off_prof; }
  void remap(size_t bytes) {
    if (base) munmap(base, mapped);
    base = nullptr;
    void *m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    if (m == MAP_FAILED) throw Error{kEIO, "shareOpacity: mmap of " + name + " failed: " + std::strerror(errno)};
    base = static_cast<char *>(m);
    mapped = bytes;
  }
  void unmap() {
    if (base) munmap(base, mapped);
    if (fd >= 0) close(fd);
    base = nullptr; fd = -1; mapped = 0;
  }
};

// Who serves `name`?  Under the name's lock: an existing segment whose owner is alive makes this
// process a client; anything else (no segment, an owner that died, a start that failed) is replaced
// by a fresh one-page segment stamped with this process's pid -- it is the owner.  Nothing else ever
// unlinks a live owner's name, and a new owner's pid is in place before the lock is released.
inline bool elect(const std::string &name, Segment &seg) {
  NameLock lock(name);
  seg.name = name;
  int fd = shm_open(name.c_str(), O_RDWR | O_CLOEXEC, 0600);
  if (fd >= 0) {
    struct stat st;
    if (fstat(fd, &st) == 0 && (size_t)st.st_size >= kPage) {
      seg.fd = fd;
      seg.remap(kPage);
      const uint32_t s = seg.hdr()->state.load(std::memory_order_acquire);
      const int owner = seg.hdr()->owner_pid.load(std::memory_order_acquire);
      if (s != kGone && seg.hdr()->version == kVersion && pid_alive(owner)) return false;   // client of `owner`
      seg.unmap();
    } else {
      close(fd);
    }
    (void)shm_unlink(name.c_str());    // stale: its owner is gone (we hold the lock: nobody is replacing it)
  }
  fd = shm_open(name.c_str(), O_CREAT | O_EXCL | O_RDWR | O_CLOEXEC, 0600);
  if (fd < 0) throw Error{kEIO, "shareOpacity: shm_open(" + name + ") failed: " + std::strerror(errno)};
  if (ftruncate(fd, kPage) != 0) { close(fd); (void)shm_unlink(name.c_str()); throw Error{kEIO, "shareOpacity: ftruncate failed"}; }
  seg.fd = fd;
  seg.remap(kPage);
  Header *h = new (seg.base) Header;
  h->state.store(kLoading);
  h->version = kVersion;
  h->maxclients = 0;
  h->nhi.store(0);
  h->total_bytes = kPage;
  h->owner_err[0] = 0;
  h->owner_pid.store((int32_t)getpid(), std::memory_order_release);
  return true;
}

// Owner, once the engine stands: grow the segment to its full size, publish the engine's facts, open for clients.
inline void publish(Segment &seg, const Info &info, int maxclients) {
  const Layout l = Layout::of(info, maxclients);
  if (ftruncate(seg.fd, (off_t)l.total) != 0) throw Error{kEIO, std::string("shareOpacity: cannot size the segment: ") + std::strerror(errno)};
  seg.remap(l.total);
  Header *h = seg.hdr();
  h->maxclients = maxclients;
  h->total_bytes = l.total;
  h->L = info.L; h->S = info.S; h->A = info.A; h->Wfull = info.Wfull; h->lo = info.lo; h->hi = info.hi;
  h->integ = info.integ; h->cut_slant = info.cut_slant; h->cia_spline = info.cia_spline; h->solution = info.solution;
  h->device = info.device;
  h->nprof = info.nprof(); h->Wl = info.Wl();
  h->off_slots = l.off_slots; h->off_prof = l.off_prof; h->off_spec = l.off_spec; h->off_over = l.off_over; h->off_ok = l.off_ok; h->off_flag = l.off_flag;
  h->off_list = l.off_list;
  h->off_info = l.off_wn;
  h->off_wn = l.off_wn; h->off_press = l.off_press; h->off_atmprof = l.off_atmprof; h->off_angles = l.off_angles;
  h->off_species = l.off_species; h->species_len = info.species.size();
  h->bell.store(0); h->asleep.store(0); h->nbatches.store(0); h->nserved.store(0); h->nfull.store(0); h->ngathered.store(0);
  for (int i = 0; i < maxclients; i++) {
    Slot *s = new (seg.slot(i)) Slot;
    s->pid.store(0); s->st.store(kIdle); s->sleeping.store(0); s->scat_flag = -1; s->rc = 0; s->err[0] = 0;
  }
  std::memcpy(seg.base + l.off_wn, info.wn_full.data(), sizeof(double) * info.wn_full.size());
  std::memcpy(seg.base + l.off_press, info.press.data(), sizeof(double) * info.press.size());
  std::memcpy(seg.base + l.off_atmprof, info.atm_prof.data(), sizeof(double) * info.atm_prof.size());
  std::memcpy(seg.base + l.off_angles, info.angles.data(), sizeof(double) * info.angles.size());
  std::memcpy(seg.base + l.off_species, info.species.c_str(), info.species.size() + 1);
}
inline void open_for_clients(Segment &seg) { seg.hdr()->state.store(kReady, std::memory_order_release); }

// Owner whose start failed, or who is leaving: the name goes (under the lock), the state says why.
inline void retire(Segment &seg, const char *why) {
  if (!seg.base) return;
  NameLock lock(seg.name);
  if (why) std::snprintf(seg.hdr()->owner_err, sizeof seg.hdr()->owner_err, "%s", why);
  seg.hdr()->state.store(kGone, std::memory_order_release);
  (void)shm_unlink(seg.name.c_str());
  lock.remove_file();
  // whoever sleeps on a slot finds out now rather than at its next timeout
  if (seg.hdr()->maxclients > 0 && seg.mapped >= seg.hdr()->total_bytes)
    for (int i = 0; i < seg.hdr()->maxclients; i++) futex_wake(&seg.slot(i)->st);
}

// ---- client ---------------------------------------------------------------------------------
struct Client {
  Segment seg;
  int slot = -1;
  Info info;
  double over[3];         // this process's set_radius / set_cloudtop / set_scattering, NaN = the cfg's
  int scat_flag = -1;
  double spin_us = 200.0;

  Client() { over[0] = over[1] = over[2] = std::nan(""); }

  // after elect() returned false, or -- the owner's own Python thread -- on the owner's mapping
  void attach(bool wait_ready) {
    const double patience = env_num("BARTRT_SHARE_LOAD_S", 600.0);
    spin_us = env_num("BARTRT_SVC_SPIN_US", 200.0);
    const auto t0 = clk::now();
    Header *h = seg.hdr();
    while (wait_ready) {
      const uint32_t s = h->state.load(std::memory_order_acquire);
      if (s == kReady) break;
      const int owner = h->owner_pid.load();
      if (s == kGone) throw Error{kENODEV, std::string("shareOpacity: the owning process (pid ") + std::to_string(owner) +
                                            ") failed to start the engine: " + h->owner_err};
      if (!pid_alive(owner)) throw Error{kENODEV, "shareOpacity: the owning process (pid " + std::to_string(owner) + ") died while loading"};
      if (since(t0) > patience) throw Error{kEIO, "shareOpacity: timed out waiting for the owning process to load the opacity grid"};
      std::this_thread::sleep_for(std::chrono::milliseconds(2));
    }
    if (seg.mapped < h->total_bytes) { seg.remap(h->total_bytes); h = seg.hdr(); }
    info.L = h->L; info.S = h->S; info.A = h->A; info.Wfull = h->Wfull; info.lo = h->lo; info.hi = h->hi;
    info.integ = h->integ; info.cut_slant = h->cut_slant; info.cia_spline = h->cia_spline; info.solution = h->solution;
    info.device = h->device;
    auto arr = [&](uint64_t off, size_t n) { const double *p = reinterpret_cast<const double *>(seg.base + off); return std::vector<double>(p, p + n); };
    info.wn_full = arr(h->off_wn, (size_t)h->Wfull);
    info.press = arr(h->off_press, (size_t)h->L);
    info.atm_prof = arr(h->off_atmprof, (size_t)h->nprof);
    info.angles = arr(h->off_angles, (size_t)h->A);
    info.species.assign(seg.base + h->off_species, h->species_len);
    for (int i = 0; i < h->maxclients; i++) {
      int32_t expect = 0;
      if (seg.slot(i)->pid.compare_exchange_strong(expect, (int32_t)getpid())) { slot = i; break; }
    }
    if (slot >= 0) {
      int32_t hi = h->nhi.load();
      while (hi < slot + 1 && !h->nhi.compare_exchange_weak(hi, slot + 1)) {}
    }
    if (slot < 0) throw Error{kENOTSUP, "shareOpacity: all " + std::to_string(h->maxclients) + " client slots of the service are taken (BARTRT_SVC_MAXCLIENTS)"};
    Slot *s = seg.slot(slot);
    s->st.store(kIdle); s->sleeping.store(0);
  }

  void check_owner() const {
    const Header *h = seg.hdr();
    const int owner = h->owner_pid.load();
    if (h->state.load(std::memory_order_acquire) != kReady || !pid_alive(owner))
      throw Error{kENODEV, "shareOpacity: the process that owns the engine (pid " + std::to_string(owner) + ") is gone"};
  }

  // trm.run_transit through the service: prof[nprof] -> spec[Wl]
  void call(const double *prof, double *spec, unsigned char *ok) {
    Header *h = seg.hdr();
    Slot *s = seg.slot(slot);
    check_owner();
    std::memcpy(seg.prof(slot), prof, sizeof(double) * (size_t)h->nprof);
    double *ov = seg.over(slot);
    ov[0] = over[0]; ov[1] = over[1]; ov[2] = over[2];
    s->scat_flag = scat_flag;
    s->st.store(kPosted);                 // (seq_cst: pairs with the dispatcher's asleep / scan)
    h->bell.fetch_add(1);
    if (h->asleep.load()) futex_wake(&h->bell);
    const auto t0 = clk::now();
    uint32_t st;
    for (;;) {
      st = s->st.load(std::memory_order_acquire);
      if (st != kPosted) break;
      if (since(t0) * 1e6 < spin_us) { cpu_relax(); continue; }
      s->sleeping.store(1);
      if (s->st.load() == kPosted) futex_wait(&s->st, kPosted, 0.05);
      s->sleeping.store(0);
      if (s->st.load(std::memory_order_acquire) == kPosted) {
        try { check_owner(); } catch (...) { s->st.store(kIdle); throw; }
      }
    }
    if (st == kFailed) {
      Error e{s->rc, s->err};
      s->st.store(kIdle);
      throw e;
    }
    std::memcpy(spec, seg.spec(slot), sizeof(double) * (size_t)h->Wl);
    const unsigned char good = *seg.ok(slot);
    s->st.store(kIdle);
    if (ok) *ok = good;
    else if (!good) throw Error{kEINVAL, "run_transit: the profile holds a non-finite or non-positive temperature"};
  }

  void detach() {
    if (slot >= 0 && seg.base) {
      seg.slot(slot)->st.store(kIdle);
      seg.slot(slot)->pid.store(0, std::memory_order_release);
    }
    slot = -1;
  }
};

// ---- dispatcher -----------------------------------------------------------------------------
// What one launch computes: the n slots of `slots` (ascending; consecutive or not -- a worker in the middle of the range
// that missed the round does not split the batch), all with the same scattering flag.  `nominal` is the number of
// walkers the KERNEL is to be chosen for: the registered clients (or BARTRT_SVC_KERNEL_WALKERS), not the n that happened
// to post together -- every kernel computes a walker independently of its neighbours, so a straggler's launch then
// gives the bits the full batch would have given it (the reference's worker calls its own engine: the same chain is
// the same bits every run, code/BARTfunc.py:363).  Throws Error / anything with what(): the slots of the run are
// failed with the message.
struct Backend {
  virtual void run(const int *slots, int n, int nominal, int scat_flag, bool any_over, bool any_cloud) = 0;
  virtual ~Backend() {}
};

struct Dispatcher {
  Segment *seg = nullptr;
  Backend *backend = nullptr;
  std::atomic<bool> stop{false};
  double window_us = 30.0, idle_spin_us = 100.0;
  bool wait_all = false;                 // BARTRT_SVC_WAIT_ALL: the window waits for every REGISTERED client, not only the active ones
  int kernel_walkers = 0;                // BARTRT_SVC_KERNEL_WALKERS: pins Backend::run's `nominal` (0: the registered clients)
  std::vector<unsigned char> active;     // slots expected in the next batch: they were in the last one (or posted since)

  void loop() {
    Header *h = seg->hdr();
    active.assign(h->maxclients, 0);
    std::vector<int> posted, group;
    auto last_prune = clk::now();
    auto idle_since = clk::now();
    while (!stop.load(std::memory_order_acquire)) {
      const uint32_t b = h->bell.load();
      const int nmax = h->nhi.load(std::memory_order_acquire);     // (slots are taken lowest first: nothing lives above)
      posted.clear();
      for (int i = 0; i < nmax; i++)
        if (seg->slot(i)->st.load(std::memory_order_acquire) == kPosted) posted.push_back(i);
      if (posted.empty()) {
        if (since(last_prune) > 0.05) { prune(); last_prune = clk::now(); }
        if (since(idle_since) * 1e6 < idle_spin_us) { cpu_relax(); continue; }
        h->asleep.store(1);
        bool any = false;
        for (int i = 0; i < nmax && !any; i++) any = seg->slot(i)->st.load() == kPosted;
        if (!any && !stop.load()) futex_wait(&h->bell, b, 0.05);
        h->asleep.store(0);
        continue;
      }
      // the window: until every active client has posted, or window_us have passed since the last arrival
      auto t_arr = clk::now();
      size_t seen = posted.size();
      for (int i : posted) active[i] = 1;
      for (;;) {
        bool all = true;
        size_t np = 0;
        for (int i = 0; i < nmax; i++) {
          const bool p = seg->slot(i)->st.load(std::memory_order_acquire) == kPosted;
          np += p;
          if (p) active[i] = 1;
          else if (active[i] || (wait_all && seg->slot(i)->pid.load(std::memory_order_relaxed) != 0)) all = false;
        }
        if (np > seen) { seen = np; t_arr = clk::now(); }
        if (all || since(t_arr) * 1e6 >= window_us) break;
        cpu_relax();
      }
      posted.clear();
      int nreg = 0;
      for (int i = 0; i < nmax; i++) {
        Slot *s = seg->slot(i);
        if (s->pid.load(std::memory_order_relaxed) != 0) nreg++;
        if (s->st.load(std::memory_order_acquire) == kPosted) posted.push_back(i);
        else active[i] = 0;          // missed this batch: not waited for until it posts again
      }
      // (counted before the callers are woken: a caller that reads the counters sees its own round)
      h->nbatches.fetch_add(1, std::memory_order_relaxed);
      h->nserved.fetch_add(posted.size(), std::memory_order_relaxed);
      if ((int)posted.size() == nreg) h->nfull.fetch_add(1, std::memory_order_relaxed);
      // the slots with one scattering flag go out as one launch, consecutive or not
      const int nominal = kernel_walkers > 0 ? kernel_walkers : nreg;
      while (!posted.empty()) {
        const int flag = seg->slot(posted[0])->scat_flag;
        group.clear();
        size_t keep = 0;
        for (int i : posted) {
          if (seg->slot(i)->scat_flag == flag) group.push_back(i);
          else posted[keep++] = i;
        }
        posted.resize(keep);
        serve(group, nominal, flag);
      }
      idle_since = clk::now();
    }
  }

  void serve(const std::vector<int> &slots, int nominal, int flag) {
    bool any_over = false, any_cloud = false;
    for (size_t k = 1; k < slots.size(); k++)
      if (slots[k] != slots[k - 1] + 1) { seg->hdr()->ngathered.fetch_add(1, std::memory_order_relaxed); break; }
    for (int i : slots) {
      const double *ov = seg->over(i);
      for (int j = 0; j < 3; j++) any_over |= ov[j] == ov[j];
      any_cloud |= ov[1] == ov[1];
    }
    int rc = 0;
    std::string msg;
    try {
      backend->run(slots.data(), (int)slots.size(), nominal < (int)slots.size() && kernel_walkers <= 0 ? (int)slots.size() : nominal,
                   flag, any_over, any_cloud);
    } catch (const Error &e) {
      rc = e.code; msg = e.msg;
    } catch (const std::exception &e) {
      rc = kEINVAL; msg = e.what();
    } catch (...) {
      rc = kENODEV; msg = "the engine failed on this batch";
    }
    for (int i : slots) {
      Slot *s = seg->slot(i);
      if (rc) {
        s->rc = rc;
        std::snprintf(s->err, sizeof s->err, "%s", msg.c_str());
      }
      s->st.store(rc ? kFailed : kDone);          // seq_cst: pairs with the client's sleeping / futex_wait
      if (s->sleeping.load()) futex_wake(&s->st);
    }
  }

  // slots whose process is gone are freed (a worker that was killed between two steps)
  void prune() {
    Header *h = seg->hdr();
    for (int i = 0; i < h->maxclients; i++) {
      Slot *s = seg->slot(i);
      const int pid = s->pid.load();
      if (pid != 0 && !pid_alive(pid)) {
        s->st.store(kIdle);
        active[i] = 0;
        s->pid.store(0);
      }
    }
  }

  int live_clients(int but_pid) const {
    int n = 0;
    for (int i = 0; i < seg->hdr()->maxclients; i++) {
      const int pid = seg->slot(i)->pid.load();
      if (pid != 0 && pid != but_pid && pid_alive(pid)) n++;
    }
    return n;
  }
};

}  // namespace svc
}  // namespace bartrt
