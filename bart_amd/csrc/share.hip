// TableShare: see share.hpp.
#include "share.hpp"

#include <hip/hip_runtime.h>

#include <atomic>
#include <new>
#include <cerrno>
#include <chrono>
#include <csignal>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "engine.hpp"
#include "svc_core.hpp"

namespace bartrt {

namespace {

constexpr uint32_t kLoading = 0, kReady = 1, kGone = 2;
constexpr uint32_t kSegVersion = 2;
constexpr int kMaxMappers = 128;

struct Seg {
  std::atomic<uint32_t> state;
  uint32_t version;
  std::atomic<int32_t> owner_pid;
  uint64_t nbytes;
  hipIpcMemHandle_t handle;
  std::atomic<int32_t> mapper[kMaxMappers];   // pids of the processes (other than the owner) that hold the mapping
};

using svc::clk;
using svc::env_num;
using svc::pid_alive;
using svc::since;

// mappers still alive (slots of dead ones are cleared: a worker that crashed never says goodbye)
int live_mappers(Seg *s) {
  int n = 0;
  for (auto &m : s->mapper) {
    const int pid = m.load();
    if (pid == 0) continue;
    if (pid_alive(pid)) n++;
    else m.store(0);
  }
  return n;
}

}  // namespace

// Who creates, replaces or joins the segment of a name is decided under that name's file lock (svc::NameLock):
// a stale segment (its owner killed) is unlinked and re-created by exactly one of the processes that find it,
// and the new owner's pid is stamped before the lock is released -- nobody can mistake a fresh segment for a
// stale one, and nobody unlinks a name it did not just examine under the lock.
TableShare *TableShare::attach(const std::string &key, size_t nbytes, const std::function<void(double *)> &fill) {
  const std::string name = svc::hashed_name("bartrt_op_", key);
  const double patience = env_num("BARTRT_SHARE_LOAD_S", 600.0);
  const auto t0 = clk::now();
  for (;;) {
    if (since(t0) > patience) throw IoError{"shareOpacity: timed out waiting for the shared opacity grid (" + name + ")"};
    int fd = -1;
    void *m = MAP_FAILED;
    bool own = false;
    try {
      svc::NameLock lock(name);
      fd = shm_open(name.c_str(), O_RDWR | O_CLOEXEC, 0600);
      if (fd >= 0) {
        struct stat st;
        if (fstat(fd, &st) == 0 && (size_t)st.st_size >= sizeof(Seg)) m = mmap(nullptr, sizeof(Seg), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        bool live = false;
        if (m != MAP_FAILED) {
          Seg *s = static_cast<Seg *>(m);
          live = s->state.load(std::memory_order_acquire) != kGone && s->version == kSegVersion && pid_alive(s->owner_pid.load());
        }
        if (!live) {
          if (m != MAP_FAILED) munmap(m, sizeof(Seg));
          m = MAP_FAILED;
          close(fd);
          fd = -1;
          (void)shm_unlink(name.c_str());
        }
      }
      if (fd < 0) {
        fd = shm_open(name.c_str(), O_CREAT | O_EXCL | O_RDWR | O_CLOEXEC, 0600);
        if (fd < 0) throw IoError{std::string("shareOpacity: shm_open failed: ") + std::strerror(errno)};
        if (ftruncate(fd, sizeof(Seg)) != 0) { close(fd); shm_unlink(name.c_str()); throw IoError{"shareOpacity: ftruncate failed"}; }
        m = mmap(nullptr, sizeof(Seg), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (m == MAP_FAILED) { close(fd); shm_unlink(name.c_str()); throw IoError{"shareOpacity: mmap failed"}; }
        Seg *s = new (m) Seg;
        s->state.store(kLoading);
        s->version = kSegVersion;
        s->nbytes = nbytes;
        for (auto &mp : s->mapper) mp.store(0);
        s->owner_pid.store((int32_t)getpid(), std::memory_order_release);
        own = true;
      }
    } catch (const svc::Error &e) {
      throw IoError{e.msg};
    }
    Seg *s = static_cast<Seg *>(m);
    if (own) {
      // ---- this process owns the grid
      auto *t = new TableShare;
      t->owner = true; t->seg = m; t->fd = fd; t->name = name;
      try {
        HIPCHK(hipMalloc(&t->ptr, nbytes));
        fill(t->ptr);
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipIpcGetMemHandle(&s->handle, t->ptr));
      } catch (...) {
        try {
          svc::NameLock lock(name);
          s->state.store(kGone, std::memory_order_release);
          shm_unlink(name.c_str());
          lock.remove_file();
        } catch (...) {}
        if (t->ptr) (void)hipFree(t->ptr);
        munmap(m, sizeof(Seg)); close(fd);
        delete t;
        throw;
      }
      s->state.store(kReady, std::memory_order_release);
      return t;
    }
    // ---- somebody else owns it: wait for the grid, then map it
    bool again = false;
    for (;;) {
      const uint32_t stt = s->state.load(std::memory_order_acquire);
      if (stt == kReady) break;
      if (stt == kGone || !pid_alive(s->owner_pid.load())) { again = true; break; }   // the election above sorts it out
      if (since(t0) > patience) { munmap(m, sizeof(Seg)); close(fd); throw IoError{"shareOpacity: timed out waiting for the shared opacity grid"}; }
      std::this_thread::sleep_for(std::chrono::milliseconds(5));
    }
    if (again) { munmap(m, sizeof(Seg)); close(fd); std::this_thread::sleep_for(std::chrono::milliseconds(2)); continue; }
    if (s->nbytes != nbytes) {
      munmap(m, sizeof(Seg)); close(fd);
      throw IoError{"shareOpacity: the shared opacity grid has another size than this configuration's (" + name + ")"};
    }
    int myslot = -1;
    for (int i = 0; i < kMaxMappers && myslot < 0; i++) {
      int32_t expect = 0;
      if (s->mapper[i].compare_exchange_strong(expect, (int32_t)getpid())) myslot = i;
    }
    if (myslot < 0) {
      (void)live_mappers(s);     // frees the slots of dead processes; one more try
      for (int i = 0; i < kMaxMappers && myslot < 0; i++) {
        int32_t expect = 0;
        if (s->mapper[i].compare_exchange_strong(expect, (int32_t)getpid())) myslot = i;
      }
      if (myslot < 0) { munmap(m, sizeof(Seg)); close(fd); throw IoError{"shareOpacity: too many processes map this grid"}; }
    }
    if (s->state.load(std::memory_order_acquire) != kReady) {
      // the owner let go between the check above and the registration: it may have freed the grid already
      s->mapper[myslot].store(0);
      munmap(m, sizeof(Seg)); close(fd);
      continue;
    }
    auto *t = new TableShare;
    t->owner = false; t->seg = m; t->fd = fd; t->name = name; t->slot = myslot;
    void *p = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&p, s->handle, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
      s->mapper[myslot].store(0);
      munmap(m, sizeof(Seg)); close(fd);
      delete t;
      throw HipError{e, "hipIpcOpenMemHandle (shareOpacity; HSA_ENABLE_IPC_MODE_LEGACY=0 in the environment?)"};
    }
    t->ptr = static_cast<double *>(p);
    return t;
  }
}

void TableShare::release() {
  Seg *s = static_cast<Seg *>(seg);
  if (!owner) {
    if (ptr) (void)hipIpcCloseMemHandle(ptr);
    if (s && slot >= 0) s->mapper[slot].store(0);
  } else {
    // new arrivals start their own grid from here on; the processes that hold this one keep it until they let
    // go.  The allocation is freed once no LIVE process maps it; if some still do when the wait runs out it is
    // NOT freed under them -- it goes with this process (and with their references) instead.
    bool free_it = true;
    if (s) {
      try {
        svc::NameLock lock(name);
        s->state.store(kGone, std::memory_order_release);
        shm_unlink(name.c_str());
        lock.remove_file();
      } catch (...) {
        s->state.store(kGone, std::memory_order_release);
        shm_unlink(name.c_str());
      }
      const double wait_s = env_num("BARTRT_SHARE_WAIT_S", 60.0);
      const auto t0 = clk::now();
      while (live_mappers(s) > 0 && since(t0) < wait_s) std::this_thread::sleep_for(std::chrono::milliseconds(5));
      if (live_mappers(s) > 0) {
        free_it = false;
        std::fprintf(stderr, "libbartrt: shareOpacity: %d process(es) still map the opacity grid after %.0f s; "
                             "it is left allocated until this process exits\n", live_mappers(s), wait_s);
      }
    }
    if (ptr && free_it) (void)hipFree(ptr);
  }
  if (seg) munmap(seg, sizeof(Seg));
  if (fd >= 0) close(fd);
  delete this;
}

}  // namespace bartrt
