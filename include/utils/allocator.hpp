// utils/allocator.hpp -- the allocation seam (reference: include/utils/allocator.hpp:11-12,
// src/allocator.cpp:8-43).  Platform::cpu gives 64-byte aligned host memory as
// before; Platform::gpu gives HBM (hipMalloc through the C ABI), which is where
// x and y must live for the timed loop to be device-resident.
#ifndef CFS_ALLOCATOR_HPP
#define CFS_ALLOCATOR_HPP

#include <cstddef>

#include "cfs_config.hpp"
#include "platform.hpp"

namespace cfs {
namespace util {
namespace memory {

void *internal_alloc(size_t bytes, Platform platform = Platform::cpu);
void internal_free(void *pointer, Platform platform = Platform::cpu);
// new in this build: move a vector between host memory and Platform::gpu memory
void internal_copy(void *dst, Platform dst_platform, const void *src, Platform src_platform,
                   size_t bytes);

} // namespace memory
} // namespace util
} // namespace cfs

#endif
