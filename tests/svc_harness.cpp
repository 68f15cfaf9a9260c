// CPU harness for the chain-service protocol (bart_amd/csrc/svc_core.hpp): the same election, slots, futexes and
// dispatcher loop as the library, with an arithmetic stand-in where the library launches the engine.  Test
// infrastructure only (tests/test_service_protocol.py builds and drives it); the product has no such backend.
//
//   svc_harness <key> <rank> <nrounds> [die_after_round]
// Every process elects on <key>; the winner owns the "engine" (a thread running svc::Dispatcher) and is a client
// like the others.  Each client posts nrounds profiles and checks what comes back; prints one JSON line.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>

#include "../bart_amd/csrc/svc_core.hpp"

using namespace bartrt::svc;

static const int kNprofL = 7, kS = 2, kW = 513;

// spec[j] = (sum of the profile) * (j + 1) + radius override (if any) + 1000 * scat_flag;  ok = prof[0] > 0
struct Echo : Backend {
  Segment *seg;
  unsigned long launches = 0;
  unsigned long gathered = 0;     // launches whose slots were not consecutive
  int nominal_min = 1 << 30, nominal_max = 0;
  void run(const int *slots, int n, int nominal, int scat_flag, bool, bool) override {
    launches++;
    for (int k = 1; k < n; k++) if (slots[k] != slots[k - 1] + 1) { gathered++; break; }
    nominal_min = nominal < nominal_min ? nominal : nominal_min;
    nominal_max = nominal > nominal_max ? nominal : nominal_max;
    const Header *h = seg->hdr();
    for (int k = 0; k < n; k++) {
      const int i = slots[k];
      const double *p = seg->prof(i);
      double s = 0;
      for (int k = 0; k < h->nprof; k++) s += p[k];
      const double *ov = seg->over(i);
      const double extra = (ov[0] == ov[0] ? ov[0] : 0.0) + 1000.0 * scat_flag;
      double *o = seg->spec(i);
      for (int j = 0; j < h->Wl; j++) o[j] = s * (j + 1) + extra;
      *seg->ok(i) = p[0] > 0;
    }
    if (std::getenv("SVC_HARNESS_FAIL_BATCH") && launches == 3) throw Error{kEINVAL, "stand-in failure of one batch"};
  }
};

int main(int argc, char **argv) {
  if (argc < 4) return 2;
  const std::string key = argv[1];
  const int rank = std::atoi(argv[2]), nrounds = std::atoi(argv[3]);
  const int die_after = argc > 4 ? std::atoi(argv[4]) : -1;
  const std::string name = hashed_name("bartrt_svctest_", key);
  Segment seg;
  Client cli;
  Dispatcher disp;
  Echo echo;
  std::thread th;
  bool owner = false;
  Segment own;
  try {
    owner = elect(name, seg);
    if (owner) {
      if (std::getenv("SVC_HARNESS_SLOW_OWNER")) std::this_thread::sleep_for(std::chrono::milliseconds(300));
      if (std::getenv("SVC_HARNESS_OWNER_FAILS")) {
        retire(seg, "stand-in: the engine did not start");
        std::printf("{\"rank\": %d, \"owner\": true, \"failed_start\": true}\n", rank);
        return 0;
      }
      Info info;
      info.L = kNprofL; info.S = kS; info.A = 1; info.Wfull = kW; info.lo = 0; info.hi = kW;
      info.wn_full.resize(kW); for (int i = 0; i < kW; i++) info.wn_full[i] = 1000.0 + i;
      info.press.assign(kNprofL, 1.0); info.atm_prof.assign((kS + 1) * kNprofL, 0.5); info.angles = {0.0};
      info.species = "H2 He";
      own = seg;
      publish(own, info, (int)env_num("BARTRT_SVC_MAXCLIENTS", 16.0));
      echo.seg = &own;
      disp.seg = &own; disp.backend = &echo;
      disp.window_us = env_num("BARTRT_SVC_WINDOW_US", 200.0);
      disp.kernel_walkers = (int)env_num("BARTRT_SVC_KERNEL_WALKERS", 0.0);
      th = std::thread([&] { disp.loop(); });
      cli.seg.name = name;
      cli.seg.fd = shm_open(name.c_str(), O_RDWR, 0600);
      cli.seg.remap(own.hdr()->total_bytes);
      cli.attach(false);
      open_for_clients(own);
    } else {
      cli.seg = seg;
      cli.attach(true);
    }
  } catch (const Error &e) {
    std::printf("{\"rank\": %d, \"owner\": %s, \"attach_error\": %d, \"msg\": \"%s\"}\n", rank, owner ? "true" : "false", e.code, e.msg.c_str());
    return 0;
  }
  const Header *h = cli.seg.hdr();
  int bad = 0, err = 0, done = 0;
  std::string msg;
  std::vector<double> prof(h->nprof), spec(h->Wl);
  if (rank % 2) cli.over[0] = 0.25 * rank;       // "set_radius" of the odd ranks only
  if (rank == 2) cli.scat_flag = 2;              // one client with another scattering flag: its own launch
  // the workers of a run start their loop together (MC3's first Scatter): wait for the announced number
  const int expect = (int)env_num("SVC_HARNESS_NCLIENTS", 1.0);
  const auto t_sync = clk::now();
  for (;;) {
    int n = 0;
    for (int i = 0; i < h->maxclients; i++) n += cli.seg.slot(i)->pid.load() != 0;
    if (n >= expect || since(t_sync) > 10.0) break;
    std::this_thread::sleep_for(std::chrono::milliseconds(1));
  }
  const auto t0 = clk::now();
  for (int r = 0; r < nrounds; r++) {
    if (r == die_after) { raise(SIGKILL); }
    for (int k = 0; k < h->nprof; k++) prof[k] = 1.0 + rank + 0.001 * r + 0.01 * k;
    if (rank == 5 && r == 1) prof[0] = -1.0;      // an unusable profile: flagged, not fatal to the batch
    unsigned char ok = 9;
    try {
      cli.call(prof.data(), spec.data(), &ok);
    } catch (const Error &e) {
      err = e.code; msg = e.msg;
      if (e.code == kENODEV) break;
      continue;
    }
    double s = 0;
    for (int k = 0; k < h->nprof; k++) s += prof[k];
    const double extra = (rank % 2 ? 0.25 * rank : 0.0) + 1000.0 * (rank == 2 ? 2 : -1);
    for (int j = 0; j < h->Wl; j++) bad += spec[j] != s * (j + 1) + extra;
    bad += ok != (prof[0] > 0 ? 1 : 0);
    done++;
    // "think time" between two calls: of every client, or (SVC_HARNESS_SLOW_RANK) of one straggler only
    if (std::getenv("SVC_HARNESS_THINK_US") && (!std::getenv("SVC_HARNESS_SLOW_RANK") || (int)env_num("SVC_HARNESS_SLOW_RANK", -1) == rank))
      std::this_thread::sleep_for(std::chrono::microseconds((long)env_num("SVC_HARNESS_THINK_US", 0)));
  }
  const double dt = since(t0);
  const int myslot = cli.slot;
  if (std::getenv("SVC_HARNESS_LEAVE_TOGETHER")) {
    // MC3 ends all chains together (code/BARTfunc.py:405-412): nobody lets go of its slot while another still runs
    const auto tl = clk::now();
    while (h->nserved.load() < (unsigned long long)expect * (unsigned long long)nrounds && since(tl) < 30.0)
      std::this_thread::sleep_for(std::chrono::milliseconds(1));
  }
  unsigned long long nb = h->nbatches.load(), ns = h->nserved.load(), nf = h->nfull.load();
  cli.detach();
  if (owner) {
    const auto tw = clk::now();
    while (disp.live_clients((int)getpid()) > 0 && since(tw) < 20.0) std::this_thread::sleep_for(std::chrono::milliseconds(2));
    nb = h->nbatches.load(); ns = h->nserved.load(); nf = h->nfull.load();   // (the whole run's, once everybody has left)
    retire(own, "released");
    disp.stop.store(true);
    futex_wake(&own.hdr()->bell);
    th.join();
  }
  std::printf("{\"rank\": %d, \"owner\": %s, \"done\": %d, \"bad\": %d, \"err\": %d, \"msg\": \"%s\", \"us_per_call\": %.2f, "
              "\"batches\": %llu, \"served\": %llu, \"full\": %llu, \"launches\": %lu, \"gathered\": %lu, "
              "\"nominal_min\": %d, \"nominal_max\": %d, \"slot\": %d}\n",
              rank, owner ? "true" : "false", done, bad, err, msg.c_str(), dt / (nrounds > 0 ? nrounds : 1) * 1e6, nb, ns, nf, echo.launches,
              echo.gathered, echo.nominal_min, echo.nominal_max, myslot);
  return 0;
}
