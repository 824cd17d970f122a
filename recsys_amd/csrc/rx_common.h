// Shared host/device helpers for librecsys_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/recsys_hip.h"

#define RX_WAVE 64
#define RX_BLOCK 256

// ---------------------------------------------------------------- host-side error plumbing
extern thread_local char rx_err_buf[512];

#define RX_FAIL(...)                                       \
  do {                                                     \
    snprintf(rx_err_buf, sizeof(rx_err_buf), __VA_ARGS__); \
    return 1;                                              \
  } while (0)

#define RX_REQUIRE(cond, ...) \
  do {                        \
    if (!(cond)) RX_FAIL(__VA_ARGS__); \
  } while (0)

#define RX_HIP(call)                                                                      \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess) RX_FAIL("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                                  __FILE__, __LINE__);                                    \
  } while (0)

// Make `device` current for the calling thread (autograd calls backward on worker threads).
#define RX_ENTER(device)                                 \
  do {                                                   \
    RX_REQUIRE((device) >= 0, "negative device ordinal"); \
    RX_HIP(hipSetDevice(device));                        \
  } while (0)

#define RX_LAUNCH_CHECK() RX_HIP(hipGetLastError())

static inline int64_t rx_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t rx_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Lanes that cooperate on one embedding row: a power of two <= 64, `vec` floats per lane per
// step.  vec = 4 (16-byte accesses) when the row width allows it, else 1.
struct RxRowShape {
  int vec;    // 1 or 4
  int group;  // lanes per row (power of two, 1..64)
  int chunks; // steps each lane takes to cover the row
};

static inline RxRowShape rx_row_shape(int dim) {
  RxRowShape s;
  s.vec = (dim % 4 == 0) ? 4 : 1;
  int units = dim / s.vec;
  int g = 1;
  while (g < units && g < 64) g <<= 1;
  s.group = g;
  s.chunks = (units + g - 1) / g;
  return s;
}

static inline bool rx_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---------------------------------------------------------------- device helpers
#ifdef __HIPCC__

__device__ __forceinline__ int rx_lane() { return threadIdx.x & 63; }

// Sum over the `group` lanes (power of two, aligned) that share a row.
__device__ __forceinline__ float rx_group_sum(float v, int group) {
  for (int m = group >> 1; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

__device__ __forceinline__ double rx_wave_sum_f64(double v) {
  for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

__device__ __forceinline__ float rx_wave_sum(float v) {
  for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// Validated row lookup: returns false (and flags) when out of range.
__device__ __forceinline__ bool rx_row_ok(int64_t row, int64_t rows, int32_t* err_flag) {
  if (row >= 0 && row < rows) return true;
  if (err_flag) atomicOr(err_flag, RX_ERR_INDEX);
  return false;
}

// VEC consecutive floats of a row held by one lane (VEC = 4: one 16-byte access).
template <int VEC>
struct RxPack {
  float v[VEC];
};

template <int VEC>
__device__ __forceinline__ RxPack<VEC> rx_load(const float* p) {
  RxPack<VEC> r;
  if constexpr (VEC == 4) {
    float4 t = *reinterpret_cast<const float4*>(p);
    r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
  } else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) r.v[i] = p[i];
  }
  return r;
}

template <int VEC>
__device__ __forceinline__ void rx_store(float* p, const RxPack<VEC>& r) {
  if constexpr (VEC == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]);
  } else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) p[i] = r.v[i];
  }
}

template <int VEC>
__device__ __forceinline__ RxPack<VEC> rx_zero() {
  RxPack<VEC> r;
#pragma unroll
  for (int i = 0; i < VEC; ++i) r.v[i] = 0.f;
  return r;
}

#endif  // __HIPCC__
