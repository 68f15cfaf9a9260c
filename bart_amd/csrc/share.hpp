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
// segment holds the pids of the mapping processes: the owner frees the allocation when it is released
// AND no live process maps it any more (it waits BARTRT_SHARE_WAIT_S, default 60 s, for that; if live
// mappers remain the allocation is left to the end of the process rather than pulled from under them).
// Creation, take-over of a dead owner's name and removal are serialised by a file lock per name.
struct TableShare {
  double *ptr = nullptr;
  bool owner = false;
  void *seg = nullptr;      // the mapped control segment
  int fd = -1;
  int slot = -1;            // this mapper's entry in the segment's pid list
  std::string name;

  static TableShare *attach(const std::string &key, size_t nbytes, const std::function<void(double *)> &fill);
  void release();           // unmap / free as described above; the object is deleted
};

}  // namespace bartrt
