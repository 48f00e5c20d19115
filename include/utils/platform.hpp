// utils/platform.hpp -- enumerations and the result-comparison helper of the
// cfs-spmv operator surface (mirrors the reference's include/utils/platform.hpp:20-37:
// same names, same enumerator values; `gpu` is appended so that cpu stays 0).
#ifndef CFS_PLATFORM_HPP
#define CFS_PLATFORM_HPP

#include <cmath>

#include "cfs_config.hpp"

namespace cfs {
namespace util {

using namespace std;

// Platform::cpu is kept for source compatibility only: this build runs the hot
// path on the GPU and refuses (loudly) a matrix explicitly created for cpu.
enum class Platform { cpu, gpu };
enum class Kernel { SpDMV };
enum class Tuning { None, Aggressive };
enum class Format { none, csr, sss, hyb };

inline int iceildiv(const int a, const int b) { return (a + b - 1) / b; }

// The reference's pass criterion (Knuth 4.2.2): |x - y| <= eps * |x| with
// eps = 1e-4 for float and 1e-8 for double.
inline bool isEqual(float x, float y) { return fabsf(x - y) <= 1e-4f * fabsf(x); }
inline bool isEqual(double x, double y) { return fabs(x - y) <= 1e-8 * fabs(x); }

} // namespace util
} // namespace cfs

#endif
