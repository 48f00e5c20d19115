// utils/runtime.hpp -- runtime knobs (reference: include/utils/runtime.hpp:15-23,
// src/runtime.cpp:10-34).  The OpenMP thread runtime of the reference is
// replaced by a HIP-stream runtime: CFS_NUM_THREADS is still parsed (the driver
// prints it) but the work is scheduled onto persistent workgroups of one GPU.
#ifndef CFS_RUNTIME_HPP
#define CFS_RUNTIME_HPP

#include <cstddef>

#include "cfs_config.hpp"

namespace cfs {
namespace util {
namespace runtime {

const int MaxThreads = 96;

// CFS_NUM_THREADS: unset -> 1, negative -> 1, "0" stays 0 (as in the reference)
size_t get_num_threads();
// threads for host-side work (parallel reader): the OpenMP default capped by the
// affinity mask and the cgroup CPU quota; CFS_HOST_THREADS overrides
int get_host_threads();
// CFS_DEVICE: HIP device ordinal this process binds to (default 0)
int get_device();
// number of HIP devices visible; 0 means the GPU path cannot run
int get_num_devices();
// CFS_NUM_GPUS: GPUs one matrix is sharded over by tune() (1-D row blocks, one HIP
// stream per device, driven by the calling thread).  Parsed like CFS_NUM_THREADS
// (src/runtime.cpp:10-21 of the reference): unset -> 1, below 1 -> 1.  More shards
// than visible devices share devices round-robin.
int get_num_gpus();
// wait for every SpMV enqueued on Platform::gpu vectors (a benchmark loop calls
// this once before it stops its clock)
void synchronize();
// kept for API compatibility (the reference never calls it either)
void setaffinity_oncpu(unsigned int cpu);

} // namespace runtime
} // namespace util
} // namespace cfs

#endif
