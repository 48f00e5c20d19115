// cfs_runtime.hpp -- the HIP-stream runtime and the memory pools behind the C ABI.
//
// What the reference has at this point: src/runtime.cpp:10-21 (CFS_NUM_THREADS is
// latched once, one OpenMP team runs every SpMV) and src/allocator.cpp:8-43
// (posix_memalign / free keyed by Platform).  Here: one context per DEVICE (its
// library stream, created on first use and never destroyed by the use of another
// device), handles that remember the device they live on, and a pool of
// page-locked host blocks for everything that crosses PCIe.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "cfs_hip.h"

namespace cfs_rt {

constexpr int kMaxDevices = 64;

inline std::string &last_error() {
  static thread_local std::string e;
  return e;
}
inline int set_err(int code, const std::string &msg) {
  last_error() = msg;
  return code;
}
#define HIPCHK(expr)                                                          \
  do {                                                                        \
    hipError_t e__ = (expr);                                                  \
    if (e__ != hipSuccess)                                                    \
      return cfs_rt::set_err(CFS_HIP_ERR_DEVICE, std::string(#expr) + ": " +  \
                                                     hipGetErrorString(e__)); \
  } while (0)

struct DevCtx {
  hipStream_t stream = nullptr; // the library stream of this device
  int num_cus = 0;
};

struct Runtime {
  std::mutex mu;
  DevCtx ctx[kMaxDevices];
  // device of the synchronous entry points (cfs_hip_sym_spmv with host pointers,
  // cfs_hip_alloc, cfs_hip_memcpy): the last cfs_hip_init(), or -- when the caller
  // never called it -- the device that was current in the calling thread at the
  // first use (torch.cuda.set_device(local_rank) of a rank process), never a
  // hard-wired device 0
  // (atomic: two threads may make the first call of a process at once; the contexts
  // themselves are created under `mu`, device_ctx)
  std::atomic<int> home{-1};
};
inline Runtime &rt() {
  static Runtime r;
  return r;
}

// make `device` current for this thread, restore the previous one on exit
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  explicit DeviceGuard(int device) {
    if (device < 0) return;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device) switched = hipSetDevice(device) == hipSuccess;
  }
  ~DeviceGuard() {
    if (switched && prev >= 0) (void)hipSetDevice(prev);
  }
};

// context of `device`; its stream is created on first use.  Other devices'
// contexts are left alone.
inline int device_ctx(int device, DevCtx **out) {
  if (device < 0 || device >= kMaxDevices) return set_err(CFS_HIP_ERR_ARG, "bad device index");
  Runtime &r = rt();
  std::lock_guard<std::mutex> lk(r.mu);
  DevCtx &c = r.ctx[device];
  if (!c.stream) {
    DeviceGuard g(device);
    // a BLOCKING stream: work of a legacy caller on the null stream (its own
    // hipMemcpy of y after an SpMV on resident vectors) orders behind it
    HIPCHK(hipStreamCreateWithFlags(&c.stream, hipStreamDefault));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    c.num_cus = prop.multiProcessorCount;
  }
  *out = &c;
  return 0;
}

inline int bind_home(int device) {
  DevCtx *c;
  HIPCHK(hipSetDevice(device));
  int rc = device_ctx(device, &c);
  if (rc) return rc;
  rt().home.store(device, std::memory_order_release);
  return 0;
}

// first use without cfs_hip_init(): adopt the caller's current device (two threads racing
// here both bind the device that is current in THEIR thread; the later store wins, both
// contexts exist)
inline int ensure_home() {
  if (rt().home.load(std::memory_order_acquire) >= 0) return 0;
  int d = 0;
  HIPCHK(hipGetDevice(&d));
  return bind_home(d);
}
inline hipStream_t home_stream() {
  const int h = rt().home.load(std::memory_order_acquire);
  if (h < 0) return nullptr;
  std::lock_guard<std::mutex> lk(rt().mu);
  return rt().ctx[h].stream;
}

// ---------------------------------------------------------------------------
// pinned-host pool (CFS_HIP_MEM_PINNED).  hipHostMalloc costs milliseconds per
// hundred megabytes, a drop-in caller allocates and frees its vectors per run:
// blocks are rounded to a power of two (>= 64 KiB), kept on a free list when
// released and handed out again.  Pointers are remembered so that free() of a
// block finds its class (and so that the allocator seam can tell a pooled block
// from a malloc'ed one).
// ---------------------------------------------------------------------------
struct PinnedPool {
  std::mutex mu;
  std::map<void *, size_t> live;               // block -> class bytes
  std::map<size_t, std::vector<void *>> spare; // class bytes -> released blocks
  size_t held = 0;                             // bytes in `spare`
  static size_t cls(size_t bytes) {
    size_t c = 64 * 1024;
    while (c < bytes) c <<= 1;
    return c;
  }
  int alloc(size_t bytes, void **out) {
    const size_t c = cls(bytes);
    {
      std::lock_guard<std::mutex> lk(mu);
      auto it = spare.find(c);
      if (it != spare.end() && !it->second.empty()) {
        *out = it->second.back();
        it->second.pop_back();
        held -= c;
        live[*out] = c;
        return 0;
      }
    }
    HIPCHK(hipHostMalloc(out, c, hipHostMallocPortable));
    std::lock_guard<std::mutex> lk(mu);
    live[*out] = c;
    return 0;
  }
  bool owns(const void *p) {
    std::lock_guard<std::mutex> lk(mu);
    return live.count(const_cast<void *>(p)) != 0;
  }
  int release(void *p) {
    size_t c = 0;
    {
      std::lock_guard<std::mutex> lk(mu);
      auto it = live.find(p);
      if (it == live.end()) return set_err(CFS_HIP_ERR_ARG, "not a block of the pinned pool");
      c = it->second;
      live.erase(it);
      // keep at most 1 GiB of released blocks around
      if (held + c <= ((size_t)1 << 30)) {
        spare[c].push_back(p);
        held += c;
        return 0;
      }
    }
    HIPCHK(hipHostFree(p));
    return 0;
  }
  void stats(size_t *nlive, size_t *nspare, size_t *spare_bytes) {
    std::lock_guard<std::mutex> lk(mu);
    *nlive = live.size();
    size_t k = 0;
    for (auto &e : spare) k += e.second.size();
    *nspare = k;
    *spare_bytes = held;
  }
};
inline PinnedPool &pinned() {
  static PinnedPool p;
  return p;
}

void parallel_copy(void *dst, const void *src, size_t bytes); // all host threads

// Pageable host memory -> device (current device), the schedule's big arrays.  A
// plain hipMemcpy of pageable memory moved 4 GB/s here; this copies 16 MiB pieces
// into two page-locked blocks of the pool with all host threads while the previous
// piece is on the bus.
inline int staged_upload(void *dst, const void *src, size_t bytes) {
  constexpr size_t kPiece = (size_t)16 << 20;
  if (bytes < 2 * kPiece || pinned().owns(src)) {
    if (bytes) HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return 0;
  }
  struct Res {
    void *pin[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    hipStream_t st = nullptr;
    ~Res() {
      if (st) (void)hipStreamSynchronize(st); // nothing may still read the blocks
      for (int k = 0; k < 2; k++) {
        if (ev[k]) (void)hipEventDestroy(ev[k]);
        if (pin[k]) (void)pinned().release(pin[k]);
      }
      if (st) (void)hipStreamDestroy(st);
    }
  } r;
  HIPCHK(hipStreamCreateWithFlags(&r.st, hipStreamNonBlocking));
  for (int k = 0; k < 2; k++) {
    int rc = pinned().alloc(kPiece, &r.pin[k]);
    if (rc) return rc;
    HIPCHK(hipEventCreateWithFlags(&r.ev[k], hipEventDisableTiming));
  }
  bool used[2] = {false, false};
  int k = 0;
  for (size_t off = 0; off < bytes; off += kPiece, k ^= 1) {
    const size_t len = bytes - off < kPiece ? bytes - off : kPiece;
    if (used[k]) HIPCHK(hipEventSynchronize(r.ev[k]));
    parallel_copy(r.pin[k], (const char *)src + off, len);
    HIPCHK(hipMemcpyAsync((char *)dst + off, r.pin[k], len, hipMemcpyHostToDevice, r.st));
    HIPCHK(hipEventRecord(r.ev[k], r.st));
    used[k] = true;
  }
  HIPCHK(hipStreamSynchronize(r.st));
  return 0;
}

// ---------------------------------------------------------------------------
// device buffer owned by a handle
// ---------------------------------------------------------------------------
struct DevBuf {
  void *p = nullptr;
  size_t bytes = 0;
  DevBuf() {}
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(DevBuf &&o) {
    if (this != &o) {
      if (p) (void)hipFree(p);
      p = o.p;
      bytes = o.bytes;
      o.p = nullptr;
      o.bytes = 0;
    }
    return *this;
  }
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  int upload(const void *src, size_t n) {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = n;
    n += 64; // padding: clamped / one-past-the-end reads of the kernels stay inside
    HIPCHK(hipMalloc(&p, n));
    // ... and read zeros: an index read there (an empty row block at the end of colind) must be a valid one
    HIPCHK(hipMemset((char *)p + bytes, 0, 64));
    return staged_upload(p, src, bytes);
  }
  int alloc(size_t n) {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = n;
    if (n == 0) n = 16;
    HIPCHK(hipMalloc(&p, n));
    return 0;
  }
};

// page-locked staging block of a handle (host-pointer callers), from the pool
struct PinBuf {
  void *p = nullptr;
  size_t bytes = 0;
  ~PinBuf() {
    if (p) (void)pinned().release(p);
  }
  int reserve(size_t n) {
    if (p && bytes >= n) return 0;
    if (p) (void)pinned().release(p);
    p = nullptr;
    bytes = 0;
    int rc = pinned().alloc(n, &p);
    if (rc) return rc;
    bytes = n;
    return 0;
  }
};

// memcpy with all host threads the caller may use (a 12 MB vector at ~10 GB/s on
// one core would cost more than its PCIe transfer)

struct PtrInfo {
  bool device = false; // device (or managed) memory
  bool pinned = false; // page-locked host memory: DMA reads / writes it directly
  int dev = -1;        // owning device of device memory
};
inline PtrInfo classify(const void *p) {
  PtrInfo r;
  hipPointerAttribute_t a;
  hipError_t e = hipPointerGetAttributes(&a, p);
  if (e != hipSuccess) {
    (void)hipGetLastError(); // unregistered host memory: clear the sticky error
    return r;
  }
  if (a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged) {
    r.device = true;
    r.dev = a.device;
  } else if (a.type == hipMemoryTypeHost) {
    r.pinned = true;
  }
  return r;
}

} // namespace cfs_rt
