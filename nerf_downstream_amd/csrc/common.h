// Shared host/device helpers for libmink_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "mink_hip.h"

namespace mink {

void set_error(const char *fmt, ...);

#define MINK_REQUIRE(cond, ...)        \
  do {                                 \
    if (!(cond)) {                     \
      mink::set_error(__VA_ARGS__);    \
      return MINK_EINVAL;              \
    }                                  \
  } while (0)

#define MINK_HIP(expr)                                                          \
  do {                                                                          \
    hipError_t e_ = (expr);                                                     \
    if (e_ != hipSuccess) {                                                     \
      mink::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),    \
                      __FILE__, __LINE__);                                      \
      return MINK_ELAUNCH;                                                      \
    }                                                                           \
  } while (0)

#define MINK_CHECK_LAUNCH() MINK_HIP(hipGetLastError())

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t align_up(int64_t a, int64_t b) { return cdiv(a, b) * b; }

constexpr uint64_t kEmptyKey = 0xFFFFFFFFFFFFFFFFull;

// ---- device helpers ---------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t k) {  // splitmix64 finaliser
  k ^= k >> 30;
  k *= 0xbf58476d1ce4e5b9ull;
  k ^= k >> 27;
  k *= 0x94d049bb133111ebull;
  k ^= k >> 31;
  return k;
}

// (b,x,y,z) -> 16|16|16|16 bit key; returns false when a field does not fit.
__device__ __forceinline__ bool pack_key(int b, int x, int y, int z, uint64_t &key) {
  const unsigned ux = (unsigned)(x + 32768), uy = (unsigned)(y + 32768), uz = (unsigned)(z + 32768);
  const bool ok = ((unsigned)b <= 65534u) && ux <= 65535u && uy <= 65535u && uz <= 65535u;
  key = ((uint64_t)(unsigned)b << 48) | ((uint64_t)(ux & 0xFFFFu) << 32) | ((uint64_t)(uy & 0xFFFFu) << 16) |
        (uint64_t)(uz & 0xFFFFu);
  return ok;
}

__device__ __forceinline__ int4 unpack_key(uint64_t key) {
  return make_int4((int)(key >> 48), (int)((key >> 32) & 0xFFFF) - 32768, (int)((key >> 16) & 0xFFFF) - 32768,
                   (int)(key & 0xFFFF) - 32768);
}

__device__ __forceinline__ int floor_to(int c, int s) {  // floor(c / s) * s, s > 0
  int q = c / s, r = c % s;
  if (r < 0) --q;
  return q * s;
}

// lanes below me that have their bit set in a wave64 ballot mask
__device__ __forceinline__ int wave_rank(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

__device__ __forceinline__ int table_find(const uint64_t *__restrict__ tkeys, const int32_t *__restrict__ tvals,
                                          uint64_t mask, uint64_t key) {
  uint64_t s = mix64(key) & mask;
  for (;;) {
    const uint64_t k = tkeys[s];
    if (k == key) return tvals[s];
    if (k == kEmptyKey) return -1;
    s = (s + 1) & mask;
  }
}

}  // namespace mink
