// Shared host-side helpers for libprl.so (error reporting, launch checks).
#pragma once

#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/prl.h"

namespace prl {

// thread-local last error message (returned by prl_last_error()).
char* error_buffer();
int set_error(int code, const char* fmt, ...);

// prl_set_tuning table (prl_api.cpp): the value of `key`, or `dflt` when unset.  One relaxed atomic load.
int64_t tuning(int key, int64_t dflt);

}  // namespace prl

#define PRL_CHECK_ARG(cond, ...)                         \
  do {                                                   \
    if (!(cond)) return prl::set_error(PRL_EINVAL, __VA_ARGS__); \
  } while (0)

#ifdef __HIPCC__
#include <hip/hip_runtime.h>

#define PRL_HIP_CHECK(expr)                                                        \
  do {                                                                             \
    hipError_t _e = (expr);                                                        \
    if (_e != hipSuccess)                                                          \
      return prl::set_error(PRL_EFAULT, "%s failed: %s (%s:%d)", #expr,            \
                            hipGetErrorString(_e), __FILE__, __LINE__);            \
  } while (0)

#define PRL_LAUNCH_CHECK(name)                                                     \
  do {                                                                             \
    hipError_t _e = hipGetLastError();                                             \
    if (_e != hipSuccess)                                                          \
      return prl::set_error(PRL_EFAULT, "launch of %s failed: %s", name,           \
                            hipGetErrorString(_e));                                \
  } while (0)

namespace prl {

constexpr int kWave = 64;  // gfx950 wavefront

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace prl
#endif  // __HIPCC__
