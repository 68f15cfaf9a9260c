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

namespace bartrt {

namespace {

constexpr uint32_t kLoading = 0, kReady = 1, kGone = 2;
constexpr uint32_t kSegVersion = 1;

struct Seg {
  std::atomic<uint32_t> state;
  uint32_t version;
  std::atomic<int32_t> owner_pid;
  std::atomic<int32_t> nmapped;    // processes other than the owner that hold the mapping
  uint64_t nbytes;
  hipIpcMemHandle_t handle;
};

std::string seg_name(const std::string &key) {
  uint64_t h = 1469598103934665603ull;          // FNV-1a
  for (unsigned char c : key) { h ^= c; h *= 1099511628211ull; }
  char buf[64];
  std::snprintf(buf, sizeof buf, "/bartrt_op_%016llx", (unsigned long long)h);
  return buf;
}

double env_seconds(const char *name, double dflt) {
  const char *e = std::getenv(name);
  return (e && *e) ? std::atof(e) : dflt;
}

bool pid_alive(int pid) { return pid > 0 && (kill(pid, 0) == 0 || errno != ESRCH); }

using clk = std::chrono::steady_clock;
double since(clk::time_point t0) { return std::chrono::duration<double>(clk::now() - t0).count(); }

}  // namespace

TableShare *TableShare::attach(const std::string &key, size_t nbytes, const std::function<void(double *)> &fill) {
  const std::string name = seg_name(key);
  const double patience = env_seconds("BARTRT_SHARE_LOAD_S", 600.0);
  const auto t0 = clk::now();
  for (;;) {
    if (since(t0) > patience) throw IoError{"shareOpacity: timed out waiting for the shared opacity grid (" + name + ")"};
    int fd = shm_open(name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd >= 0) {
      // ---- this process owns the grid
      if (ftruncate(fd, sizeof(Seg)) != 0) { close(fd); shm_unlink(name.c_str()); throw IoError{"shareOpacity: ftruncate failed"}; }
      void *m = mmap(nullptr, sizeof(Seg), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
      if (m == MAP_FAILED) { close(fd); shm_unlink(name.c_str()); throw IoError{"shareOpacity: mmap failed"}; }
      Seg *s = new (m) Seg;
      s->state.store(kLoading);
      s->version = kSegVersion;
      s->nmapped.store(0);
      s->nbytes = nbytes;
      s->owner_pid.store((int32_t)getpid(), std::memory_order_release);
      auto *t = new TableShare;
      t->owner = true; t->seg = m; t->fd = fd; t->name = name;
      try {
        HIPCHK(hipMalloc(&t->ptr, nbytes));
        fill(t->ptr);
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipIpcGetMemHandle(&s->handle, t->ptr));
      } catch (...) {
        s->state.store(kGone, std::memory_order_release);
        shm_unlink(name.c_str());
        if (t->ptr) (void)hipFree(t->ptr);
        munmap(m, sizeof(Seg)); close(fd);
        delete t;
        throw;
      }
      s->state.store(kReady, std::memory_order_release);
      return t;
    }
    if (errno != EEXIST) throw IoError{std::string("shareOpacity: shm_open failed: ") + std::strerror(errno)};
    // ---- somebody else created the segment: wait for the grid, then map it
    fd = shm_open(name.c_str(), O_RDWR, 0600);
    if (fd < 0) continue;                       // unlinked in between: start over
    struct stat st;
    if (fstat(fd, &st) != 0 || (size_t)st.st_size < sizeof(Seg)) {   // not sized yet
      close(fd);
      std::this_thread::sleep_for(std::chrono::milliseconds(2));
      continue;
    }
    void *m = mmap(nullptr, sizeof(Seg), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    if (m == MAP_FAILED) { close(fd); throw IoError{"shareOpacity: mmap failed"}; }
    Seg *s = static_cast<Seg *>(m);
    bool retry = false;
    for (;;) {
      const uint32_t stt = s->state.load(std::memory_order_acquire);
      const int owner = s->owner_pid.load(std::memory_order_acquire);
      if (stt == kReady && pid_alive(owner)) break;
      const bool dead = stt == kGone || (owner != 0 && !pid_alive(owner)) || (owner == 0 && since(t0) > 5.0);
      if (dead) {
        // the owner went away (or never came up): the name is stale -- whoever gets there first replaces it
        shm_unlink(name.c_str());
        retry = true;
        break;
      }
      if (since(t0) > patience) { munmap(m, sizeof(Seg)); close(fd); throw IoError{"shareOpacity: timed out waiting for the shared opacity grid"}; }
      std::this_thread::sleep_for(std::chrono::milliseconds(5));
    }
    if (retry) { munmap(m, sizeof(Seg)); close(fd); continue; }
    if (s->version != kSegVersion || s->nbytes != nbytes) {
      munmap(m, sizeof(Seg)); close(fd);
      throw IoError{"shareOpacity: the shared opacity grid has another size than this configuration's (" + name + ")"};
    }
    s->nmapped.fetch_add(1);
    if (s->state.load(std::memory_order_acquire) != kReady) {
      // the owner let go between the check above and the count: it may have freed the grid already
      s->nmapped.fetch_sub(1);
      munmap(m, sizeof(Seg)); close(fd);
      continue;
    }
    auto *t = new TableShare;
    t->owner = false; t->seg = m; t->fd = fd; t->name = name;
    void *p = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&p, s->handle, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
      s->nmapped.fetch_sub(1);
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
    if (s) s->nmapped.fetch_sub(1);
  } else {
    // new arrivals start their own grid from here on; the processes that hold this one keep it
    // until they let go (bounded wait), then the allocation is freed
    shm_unlink(name.c_str());
    if (s) {
      s->state.store(kGone, std::memory_order_release);
      const double wait_s = env_seconds("BARTRT_SHARE_WAIT_S", 60.0);
      const auto t0 = clk::now();
      while (s->nmapped.load() > 0 && since(t0) < wait_s) std::this_thread::sleep_for(std::chrono::milliseconds(5));
    }
    if (ptr) (void)hipFree(ptr);
  }
  if (seg) munmap(seg, sizeof(Seg));
  if (fd >= 0) close(fd);
  delete this;
}

}  // namespace bartrt
