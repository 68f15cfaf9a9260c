// ChainService: see svc.hpp / svc_core.hpp.
#include "svc.hpp"

#include <hip/hip_runtime.h>

#include "engine.hpp"

namespace bartrt {

// A round whose slots are not consecutive (a worker in the middle of the range missed it) still is ONE launch: the
// posted profiles and overrides are gathered from their slots into the engine's own buffers, the batch runs on those,
// and the spectra and flags are scattered back to the slots -- two small launches (a few us each) instead of a second RT
// launch.  `list` (the slots) lies in the segment itself, like everything these kernels touch on the host side.
__global__ void svc_gather(const double *__restrict__ seg_prof, const double *__restrict__ seg_over,
                           const int32_t *__restrict__ list, int nprof, double *__restrict__ prof, double *__restrict__ over) {
  const int w = blockIdx.x, s = list[w];
  const double *src = seg_prof + (size_t)s * nprof;
  double *dst = prof + (size_t)w * nprof;
  for (int i = threadIdx.x; i < nprof; i += blockDim.x) dst[i] = src[i];
  if (threadIdx.x < 3) over[3 * w + threadIdx.x] = seg_over[3 * (size_t)s + threadIdx.x];
}
__global__ void svc_scatter(const double *__restrict__ spec, const unsigned char *__restrict__ ok,
                            const int32_t *__restrict__ list, int Wl, double *__restrict__ seg_spec,
                            unsigned char *__restrict__ seg_ok) {
  const int w = blockIdx.y, s = list[w];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < Wl) seg_spec[(size_t)s * Wl + i] = spec[(size_t)w * Wl + i];
  if (i == 0) seg_ok[s] = ok[w];
}

ChainService *ChainService::start(svc::Segment &&elected, Engine *e) {
  auto *s = new ChainService;
  s->seg = elected;
  elected = svc::Segment{};
  s->eng = e;
  try {
    svc::Info info;
    info.L = e->L; info.S = e->S; info.A = e->A; info.Wfull = e->Wfull; info.lo = e->lo; info.hi = e->hi;
    info.integ = e->integ; info.cut_slant = e->cut_slant ? 1 : 0; info.cia_spline = e->cia_spline ? 1 : 0;
    info.solution = e->solution; info.device = e->device;
    info.wn_full = e->wn_full;
    info.press = e->atm.press;
    info.angles = e->angles;
    info.atm_prof.resize((size_t)(e->S + 1) * e->L);
    for (int l = 0; l < e->L; l++) {
      info.atm_prof[l] = e->atm.temp[l];
      for (int k = 0; k < e->S; k++) info.atm_prof[(size_t)(k + 1) * e->L + l] = e->atm.abund[(size_t)l * e->S + k];
    }
    for (auto &n : e->atm.species) info.species += (info.species.empty() ? "" : " ") + n;
    int maxc = (int)svc::env_num("BARTRT_SVC_MAXCLIENTS", (double)svc::kDefaultClients);
    if (maxc < 1) maxc = 1;
    if (maxc > svc::kMaxClients) maxc = svc::kMaxClients;
    svc::publish(s->seg, info, maxc);
    e->ensure_walkers(maxc);
    // The kernels read the posted profiles from, and (small launches) write the spectra to, the segment itself:
    // its data area is page-locked and mapped on the device.  Where the driver refuses (it has not, on the
    // boxes measured), the batch is staged through the engine's own buffers with plain copies.
    hipError_t er = svc::env_num("BARTRT_SVC_REGISTER", 1.0) != 0.0
                        ? hipHostRegister(s->seg.data_begin(), s->seg.data_bytes(), hipHostRegisterMapped | hipHostRegisterPortable)
                        : hipErrorNotSupported;
    if (er == hipSuccess) {
      void *dev = nullptr;
      er = hipHostGetDevicePointer(&dev, s->seg.data_begin(), 0);
      if (er == hipSuccess) {
        char *d0 = static_cast<char *>(dev);
        const svc::Header *h = s->seg.hdr();
        s->d_prof = reinterpret_cast<double *>(d0);      // (the registered range starts at the profiles)
        s->d_spec = reinterpret_cast<double *>(d0 + (h->off_spec - h->off_prof));
        s->d_over = reinterpret_cast<double *>(d0 + (h->off_over - h->off_prof));
        s->d_ok = reinterpret_cast<unsigned char *>(d0 + (h->off_ok - h->off_prof));
        s->d_flag = reinterpret_cast<uint32_t *>(d0 + (h->off_flag - h->off_prof));
        s->d_list = reinterpret_cast<int32_t *>(d0 + (h->off_list - h->off_prof));
        s->registered = true;
      } else {
        (void)hipHostUnregister(s->seg.data_begin());
      }
    }
    if (!s->registered) (void)hipGetLastError();
    HIPCHK(hipMalloc(&s->d_over_stage, sizeof(double) * 3 * (size_t)maxc));   // (overrides of a gathered / staged batch)
    // spectra up to this many bytes per launch go to host memory straight from the RT kernel; above, the
    // kernel writes HBM and one DMA copy follows (measured: DESIGN.md 6, "chain service")
    s->direct_spec_bytes = (size_t)svc::env_num("BARTRT_SVC_DIRECT_BYTES", 4.0 * 1024 * 1024);
    s->sync_mode = (int)svc::env_num("BARTRT_SVC_SYNC", 1.0);   // (measured at ten clients: 121 us per call against 128 with hipStreamSynchronize, 130 with an event)
    if (s->sync_mode == 1 && !s->registered) s->sync_mode = 0;
    if (s->sync_mode == 2) {
      hipEvent_t ev;
      HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
      s->ev_done = ev;
    }
    *reinterpret_cast<volatile uint32_t *>(s->seg.base + s->seg.hdr()->off_flag) = 0;
    s->disp.seg = &s->seg;
    s->disp.backend = s;
    s->disp.window_us = svc::env_num("BARTRT_SVC_WINDOW_US", 30.0);
    s->disp.wait_all = svc::env_num("BARTRT_SVC_WAIT_ALL", 0.0) != 0.0;   // (whole batches whatever the processes' pace)
    s->disp.kernel_walkers = (int)svc::env_num("BARTRT_SVC_KERNEL_WALKERS", 0.0);
    s->disp.idle_spin_us = svc::env_num("BARTRT_SVC_IDLE_SPIN_US", 100.0);
    const int device = e->device;
    s->th = std::thread([s, device] {
      (void)hipSetDevice(device);
      s->disp.loop();
    });
  } catch (...) {
    if (s->registered) (void)hipHostUnregister(s->seg.data_begin());
    if (s->d_over_stage) (void)hipFree(s->d_over_stage);
    elected = s->seg;          // the caller retires the name
    s->seg = svc::Segment{};
    delete s;
    throw;
  }
  return s;
}

void ChainService::wait_done() {
  Engine *e = eng;
  if (sync_mode == 1) {
    const uint32_t want = ++flag_seq;
    HIPCHK(hipStreamWriteValue32(e->stream, d_flag, want, 0));
    volatile uint32_t *f = reinterpret_cast<volatile uint32_t *>(seg.base + seg.hdr()->off_flag);
    const auto t0 = svc::clk::now();
    long spins = 0;
    while (*f != want) {
      svc::cpu_relax();
      // (a failed launch never writes: after a generous wait the stream itself is asked)
      if ((++spins & 0xffff) == 0 && svc::since(t0) > 2.0) { HIPCHK(hipStreamSynchronize(e->stream)); break; }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
  } else if (sync_mode == 2) {
    hipEvent_t ev = static_cast<hipEvent_t>(ev_done);
    HIPCHK(hipEventRecord(ev, e->stream));
    hipError_t q;
    while ((q = hipEventQuery(ev)) == hipErrorNotReady) svc::cpu_relax();
    HIPCHK(q);
  } else {
    HIPCHK(hipStreamSynchronize(e->stream));
  }
}

void ChainService::run(const int *slots, int n, int nominal, int scat_flag, bool any_over, bool any_cloud) {
  Engine *e = eng;
  const svc::Header *h = seg.hdr();
  const size_t nprof = (size_t)h->nprof, Wl = (size_t)h->Wl;
  const int flag0 = e->scat_flag;
  struct Restore { Engine *e; int f; ~Restore() { e->scat_flag = f; e->prep_over_once = nullptr; e->sel_walkers = 0; } } restore{e, flag0};
  if (scat_flag >= 0) e->scat_flag = scat_flag;
  e->ensure_walkers(n);
  e->last_prof = nullptr;
  e->last_n = 0;
  e->prep_over_cloud = any_cloud;
  // the kernel of the full batch whatever posted together (svc_core.hpp, Backend::run)
  e->sel_walkers = nominal > n ? nominal : n;
  const int first = slots[0];
  bool consecutive = true;
  for (int k = 1; k < n; k++) consecutive &= slots[k] == slots[k - 1] + 1;
  if (registered && consecutive) {
    e->prep_over_once = any_over ? d_over + 3 * (size_t)first : nullptr;
    const bool direct = sizeof(double) * (size_t)n * Wl <= direct_spec_bytes;
    e->run_dev(d_prof + (size_t)first * nprof, n, direct ? d_spec + (size_t)first * Wl : e->d_spec, d_ok + first, e->stream, false);
    if (!direct)
      HIPCHK(hipMemcpyAsync(seg.spec(first), e->d_spec, sizeof(double) * (size_t)n * Wl, hipMemcpyDeviceToHost, e->stream));
    wait_done();
  } else if (registered) {
    std::memcpy(seg.list(), slots, sizeof(int32_t) * (size_t)n);
    hipLaunchKernelGGL(svc_gather, dim3(n), dim3(256), 0, e->stream, d_prof, d_over, d_list, (int)nprof, e->d_prof, d_over_stage);
    e->prep_over_once = any_over ? d_over_stage : nullptr;
    e->run_dev(e->d_prof, n, e->d_spec, e->d_ok, e->stream, false);
    hipLaunchKernelGGL(svc_scatter, dim3((unsigned)((Wl + 255) / 256), n), dim3(256), 0, e->stream, e->d_spec, e->d_ok, d_list, (int)Wl,
                       d_spec, d_ok);
    HIPCHK(hipGetLastError());
    wait_done();
  } else {
    // (the driver refused to register the segment: plain copies, run of consecutive slots by run)
    auto runs = [&](auto &&f) {
      for (int k = 0; k < n;) {
        int e2 = k + 1;
        while (e2 < n && slots[e2] == slots[e2 - 1] + 1) e2++;
        f(k, slots[k], e2 - k);
        k = e2;
      }
    };
    runs([&](int k, int s0, int cnt) {
      HIPCHK(hipMemcpyAsync(e->d_prof + (size_t)k * nprof, seg.prof(s0), sizeof(double) * (size_t)cnt * nprof, hipMemcpyHostToDevice, e->stream));
      if (any_over) HIPCHK(hipMemcpyAsync(d_over_stage + 3 * (size_t)k, seg.over(s0), sizeof(double) * 3 * (size_t)cnt, hipMemcpyHostToDevice, e->stream));
    });
    if (any_over) e->prep_over_once = d_over_stage;
    e->run_dev(e->d_prof, n, e->d_spec, e->d_ok, e->stream, false);
    runs([&](int k, int s0, int cnt) {
      HIPCHK(hipMemcpyAsync(seg.spec(s0), e->d_spec + (size_t)k * Wl, sizeof(double) * (size_t)cnt * Wl, hipMemcpyDeviceToHost, e->stream));
      HIPCHK(hipMemcpyAsync(seg.ok(s0), e->d_ok + k, (size_t)cnt, hipMemcpyDeviceToHost, e->stream));
    });
    HIPCHK(hipStreamSynchronize(e->stream));
  }
}

void ChainService::shutdown(double wait_s) {
  // the other workers finish their run first: MC3 ends all chains together (code/BARTfunc.py:405-412), the
  // owner may simply be the first to get there
  const auto t0 = svc::clk::now();
  const int me = (int)getpid();
  while (disp.live_clients(me) > 0 && svc::since(t0) < wait_s) std::this_thread::sleep_for(std::chrono::milliseconds(2));
  svc::retire(seg, "the owning process released the engine");
  disp.stop.store(true, std::memory_order_release);
  svc::futex_wake(&seg.hdr()->bell);
  if (th.joinable()) th.join();
  // a client that posted while the dispatcher was stopping is told so (it would time out on its own otherwise)
  for (int i = 0; i < seg.hdr()->maxclients; i++) {
    svc::Slot *s = seg.slot(i);
    if (s->st.load() == svc::kPosted) {
      s->rc = svc::kENODEV;
      std::snprintf(s->err, sizeof s->err, "shareOpacity: the process that owns the engine released it");
      s->st.store(svc::kFailed);
      svc::futex_wake(&s->st);
    }
  }
  (void)hipSetDevice(eng->device);
  (void)hipDeviceSynchronize();
  if (registered) (void)hipHostUnregister(seg.data_begin());
  if (d_over_stage) (void)hipFree(d_over_stage);
  if (ev_done) (void)hipEventDestroy(static_cast<hipEvent_t>(ev_done));
  delete eng;
  eng = nullptr;
  seg.unmap();
  delete this;
}

}  // namespace bartrt
