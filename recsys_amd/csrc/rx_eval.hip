// Per-user top-k over list-wise scores (gfx950): one wavefront per user row, candidates in
// registers, k rounds of a 64-lane arg-max by (score, column).
#include "rx_common.h"

#define TOPK_PER_LANE 16  // up to 1024 candidates per row

__global__ void __launch_bounds__(RX_BLOCK)
topk_kernel(const float* __restrict__ scores, const int64_t* __restrict__ codes, int64_t rows, int n, int k,
            int64_t* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (RX_BLOCK / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* s = scores + row * n;
  float v[TOPK_PER_LANE];
  unsigned taken = 0;
#pragma unroll
  for (int q = 0; q < TOPK_PER_LANE; ++q) {
    const int c = q * 64 + lane;
    v[q] = c < n ? s[c] : 0.f;
    if (c >= n) taken |= 1u << q;
  }
  for (int r = 0; r < k; ++r) {
    // best live candidate of this lane: highest score, larger column on ties
    float bv = 0.f;
    int bc = -1;
#pragma unroll
    for (int q = 0; q < TOPK_PER_LANE; ++q) {
      const int c = q * 64 + lane;
      const bool live = !((taken >> q) & 1u);
      if (live && (bc < 0 || v[q] > bv || (v[q] == bv && c > bc))) { bv = v[q]; bc = c; }
    }
    // wave arg-max
    for (int m = 32; m > 0; m >>= 1) {
      const float ov = __shfl_xor(bv, m, 64);
      const int oc = __shfl_xor(bc, m, 64);
      if (oc >= 0 && (bc < 0 || ov > bv || (ov == bv && oc > bc))) { bv = ov; bc = oc; }
    }
    if (bc >= 0 && (bc & 63) == lane) taken |= 1u << (bc >> 6);
    if (lane == 0) out[row * k + r] = bc < 0 ? -1 : (codes ? codes[row * n + bc] : (int64_t)bc);
  }
}

extern "C" int rx_topk_codes(const float* d_scores, const int64_t* d_codes, int64_t rows, int32_t n,
                             int32_t k, int64_t* d_out, int device, void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(rows >= 0 && n >= 1 && k >= 1 && k <= n, "rx_topk_codes: bad shape rows=%lld n=%d k=%d", (long long)rows, n, k);
  RX_REQUIRE(n <= 64 * TOPK_PER_LANE, "rx_topk_codes: n=%d above %d candidates per row", n, 64 * TOPK_PER_LANE);
  if (rows == 0) return 0;
  RX_REQUIRE(d_scores && d_out, "rx_topk_codes: null argument");
  const unsigned grid = (unsigned)rx_cdiv(rows, RX_BLOCK / 64);
  hipLaunchKernelGGL(topk_kernel, dim3(grid), dim3(RX_BLOCK), 0, (hipStream_t)stream, d_scores, d_codes, rows, n, k, d_out);
  RX_LAUNCH_CHECK();
  return 0;
}
