// runtime.cpp -- environment knobs (reference: src/runtime.cpp:10-34).
#include "utils/runtime.hpp"

#include <omp.h>
#include <sched.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
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

int get_host_threads() {
  static int cached = 0;
  if (cached > 0) return cached;
  long t = omp_get_max_threads();
  cpu_set_t set;
  CPU_ZERO(&set);
  if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0 && CPU_COUNT(&set) < t)
    t = CPU_COUNT(&set);
  // a container may show every core of the host and still grant only a quota
  if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char q[64];
    long period = 0;
    if (fscanf(f, "%63s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
      const long quota = atol(q);
      if (quota > 0 && (quota + period - 1) / period < t) t = (quota + period - 1) / period;
    }
    fclose(f);
  }
  const char *env = getenv("CFS_HOST_THREADS");
  if (env && atoi(env) > 0) t = atoi(env);
  cached = t < 1 ? 1 : (int)t;
  return cached;
}

int get_device() {
  const char *env = getenv("CFS_DEVICE");
  int d = env ? atoi(env) : 0;
  return d < 0 ? 0 : d;
}

int get_num_gpus() {
  const char *env = getenv("CFS_NUM_GPUS");
  int ret = 1;
  if (env) {
    ret = atoi(env);
    if (ret < 1) ret = 1;
    if (ret > 64) ret = 64;
  }
  return ret;
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
