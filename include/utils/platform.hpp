// utils/platform.hpp -- the enumerations of the operator surface and the
// comparison helper of the self-check.  Names and enumerator values are the
// reference's (include/utils/platform.hpp:20-37); Platform::gpu is appended so
// that cpu keeps the value 0.
#ifndef CFS_PLATFORM_HPP
#define CFS_PLATFORM_HPP

#include <cmath>

#include "cfs_config.hpp"

namespace cfs {
namespace util {

using namespace std;

// where a matrix / a vector lives.  cpu: host memory (vectors) -- a MATRIX created
// for cpu is refused, this build has no CPU kernels.  gpu: HBM of the bound MI355X.
enum class Platform { cpu, gpu };

// the one kernel of the library: sparse matrix times dense vector
enum class Kernel { SpDMV };

// None: one schedule build.  Aggressive (default): also the measured steps of
// tune() -- window shape and per-XCD work shares (csr.cpp, cfs_hip.h).
enum class Tuning { None, Aggressive };

// storage asked for at create(): csr = every stored entry; sss = symmetric, lower
// triangle + diagonal; hyb = sss whose single-use halo entries leave the tile format
// (far entries, CFS_HIP_FLAG_HYB); none = unset
enum class Format { none, csr, sss, hyb };

// ceil(a / b) for positive ints (partition sizes)
inline int iceildiv(const int a, const int b) { return (a + b - 1) / b; }

// Pass criterion of the self-check, the reference's (Knuth 4.2.2, relative to the
// first argument): eps = 1e-4 in single, 1e-8 in double precision.
inline bool isEqual(float x, float y) { return fabsf(x - y) <= 1e-4f * fabsf(x); }
inline bool isEqual(double x, double y) { return fabs(x - y) <= 1e-8 * fabs(x); }

} // namespace util
} // namespace cfs

#endif
