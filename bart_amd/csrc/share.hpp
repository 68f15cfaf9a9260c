// The opacity grid of one GPU shared between the worker processes of a retrieval (the reference's
// `shareOpacity`: code/makecfg.py:106-107, BART.py:259-262 -- there a System V segment in host
// memory; here one allocation in HBM that the other processes map through HIP's IPC handles).
#pragma once
#include <cstddef>
#include <functional>
#include <string>

namespace bartrt {

// One process (the first to ask for `key`) allocates `nbytes` on the current device, fills them
// through `fill` and publishes the allocation's IPC handle in a small POSIX shared-memory segment
// named after the key; the others wait for it and map the same memory (read-only use).  The
// segment holds a count of the mappings: the owner frees the allocation when it is released AND
// the count has dropped to zero (or after a timeout, BARTRT_SHARE_WAIT_S, default 60 s).
struct TableShare {
  double *ptr = nullptr;
  bool owner = false;
  void *seg = nullptr;      // the mapped control segment
  int fd = -1;
  std::string name;

  static TableShare *attach(const std::string &key, size_t nbytes, const std::function<void(double *)> &fill);
  void release();           // unmap / free as described above; the object is deleted
};

}  // namespace bartrt
