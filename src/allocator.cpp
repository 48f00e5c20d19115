// allocator.cpp -- internal_alloc / internal_free / internal_copy
// (reference seam: src/allocator.cpp:8-43).  gpu = HBM through the C ABI.  cpu =
// 64-byte aligned host memory as in the reference; vectors of 64 KiB .. 256 MiB
// come from the page-locked pool of the runtime (CFS_HIP_MEM_PINNED) once a
// device is bound (cfs_hip_runtime_bound: creating a matrix binds it; an allocation
// alone never starts HIP), so that an unmodified caller that hands its host x / y to
// SpDMV (test/test_spmv_mmf.cpp:82-83) is DMA-ed in place instead of being
// copied through a staging block.  Failures print and exit(1), like the
// reference does.
#include "utils/allocator.hpp"

#include <cstdlib>
#include <cstring>
#include <iostream>

#include "cfs_hip.h"

namespace cfs {
namespace util {
namespace memory {

static const size_t kPinnedMin = (size_t)64 << 10, kPinnedMax = (size_t)256 << 20;

static void die(const char *what) {
  std::cout << "[ERROR]: " << what << ": " << cfs_hip_last_error() << std::endl;
  exit(1);
}

void *internal_alloc(size_t bytes, Platform platform) {
  void *pointer = nullptr;
  if (platform == Platform::gpu) {
    if (cfs_hip_alloc(bytes, CFS_HIP_MEM_DEVICE, &pointer) != 0) die("cfs_hip_alloc() failed");
    return pointer;
  }
  if (bytes >= kPinnedMin && bytes <= kPinnedMax && getenv("CFS_NO_PINNED") == nullptr &&
      cfs_hip_runtime_bound() && cfs_hip_alloc(bytes, CFS_HIP_MEM_PINNED, &pointer) == 0)
    return pointer;
  pointer = nullptr;
  if (posix_memalign(&pointer, 64, bytes ? bytes : 64) != 0) {
    std::cout << "[ERROR]: posix_memalign() failed!" << std::endl;
    exit(1);
  }
  return pointer;
}

void internal_free(void *pointer, Platform platform) {
  if (!pointer) return;
  if (platform == Platform::gpu) {
    if (cfs_hip_free(pointer, CFS_HIP_MEM_DEVICE) != 0) die("cfs_hip_free() failed");
    return;
  }
  if (cfs_hip_pinned_owns(pointer)) {
    if (cfs_hip_free(pointer, CFS_HIP_MEM_PINNED) != 0) die("cfs_hip_free() failed");
    return;
  }
  free(pointer);
}

void internal_copy(void *dst, Platform dst_platform, const void *src, Platform src_platform,
                   size_t bytes) {
  int dir = CFS_HIP_D2D;
  if (dst_platform == Platform::gpu && src_platform == Platform::cpu) dir = CFS_HIP_H2D;
  else if (dst_platform == Platform::cpu && src_platform == Platform::gpu) dir = CFS_HIP_D2H;
  else if (dst_platform == Platform::cpu && src_platform == Platform::cpu) {
    if (bytes) memcpy(dst, src, bytes);
    return;
  }
  if (cfs_hip_memcpy(dst, src, bytes, dir) != 0) die("cfs_hip_memcpy() failed");
}

} // namespace memory
} // namespace util
} // namespace cfs
