// runtime.cpp -- environment knobs (reference: src/runtime.cpp:10-34).
#include "utils/runtime.hpp"

#include <sched.h>

#include <cstdlib>
#include <iostream>

#include "cfs_hip.h"

namespace cfs {
namespace util {
namespace runtime {

size_t get_num_threads() {
  const char *env = getenv("CFS_NUM_THREADS");
  int ret = 1;
  if (env) {
    ret = atoi(env);
    if (ret < 0) ret = 1;
  }
  return ret;
}

int get_device() {
  const char *env = getenv("CFS_DEVICE");
  int d = env ? atoi(env) : 0;
  return d < 0 ? 0 : d;
}

int get_num_devices() {
  int n = 0;
  if (cfs_hip_device_count(&n) != 0) return 0;
  return n;
}

void synchronize() {
  if (cfs_hip_synchronize(nullptr) != 0) {
    std::cout << "[ERROR]: " << cfs_hip_last_error() << std::endl;
    exit(1);
  }
}

void setaffinity_oncpu(unsigned int cpu) {
  cpu_set_t mask;
  CPU_ZERO(&mask);
  CPU_SET(cpu, &mask);
  if (sched_setaffinity(0, sizeof(cpu_set_t), &mask)) {
    std::cout << "sched_setaffinity() failed" << std::endl;
    exit(1);
  }
}

} // namespace runtime
} // namespace util
} // namespace cfs
