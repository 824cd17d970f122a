// Negative sampling for an epoch of implicit-feedback training (gfx950): for every positive
// (user, item), n_neg items drawn uniformly with replacement from the user's pool, written next to
// the positive as one [1 + n_neg] row of item codes, with the matching [1, 0, ..., 0] label row.
// The random stream is counter based: draw j of positive p is a pure function of (seed, p, j), so an
// epoch is reproducible from its seed whatever the launch geometry, and a host mirror of the mixing
// function reproduces it bit for bit.
#include "rx_common.h"

// splitmix64 finaliser over a Weyl-sequence position: passes through every 64-bit value once per seed
__device__ __forceinline__ uint64_t sample_mix(uint64_t seed, uint64_t counter) {
  uint64_t z = seed + (counter + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__global__ void __launch_bounds__(RX_BLOCK)
sample_negatives_kernel(const int64_t* __restrict__ users, const int64_t* __restrict__ items,
                        const int64_t* __restrict__ pool, int64_t n, int64_t n_users, int32_t pool_size,
                        int32_t n_neg, uint64_t seed, int64_t* __restrict__ out_items,
                        double* __restrict__ out_labels, int32_t* __restrict__ err_flag) {
  const int width = 1 + n_neg;
  const int64_t total = n * width;
  for (int64_t t = (int64_t)blockIdx.x * RX_BLOCK + threadIdx.x; t < total; t += (int64_t)gridDim.x * RX_BLOCK) {
    const int64_t p = t / width;
    const int j = (int)(t - p * width);
    int64_t code;
    if (j == 0) {
      code = items[p];
    } else {
      const int64_t u = users[p];
      if (!rx_row_ok(u, n_users, err_flag)) {
        code = 0;
      } else {
        // high 32 bits of the mix, scaled to [0, pool_size): multiply-shift, bias below pool_size / 2^32
        const uint64_t r = sample_mix(seed, (uint64_t)p * (uint64_t)n_neg + (uint64_t)(j - 1)) >> 32;
        code = pool[u * pool_size + (int64_t)((r * (uint64_t)pool_size) >> 32)];
      }
    }
    out_items[t] = code;
    if (out_labels) out_labels[t] = j == 0 ? 1.0 : 0.0;
  }
}

extern "C" int rx_sample_negatives(const int64_t* d_users, const int64_t* d_items, const int64_t* d_pool,
                                   int64_t n, int64_t n_users, int32_t pool_size, int32_t n_neg, uint64_t seed,
                                   int64_t* d_out_items, double* d_out_labels, int32_t* err_flag, int device,
                                   void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(n >= 0 && n_users >= 1 && pool_size >= 1 && n_neg >= 0 && n_neg <= 4096,
             "rx_sample_negatives: bad shape n=%lld n_users=%lld pool=%d n_neg=%d", (long long)n, (long long)n_users, pool_size, n_neg);
  if (n == 0) return 0;
  RX_REQUIRE(d_users && d_items && d_pool && d_out_items, "rx_sample_negatives: null argument");
  int64_t g = rx_cdiv(n * (1 + n_neg), RX_BLOCK);
  const unsigned grid = (unsigned)(g < 8192 ? g : 8192);
  hipLaunchKernelGGL(sample_negatives_kernel, dim3(grid), dim3(RX_BLOCK), 0, (hipStream_t)stream, d_users, d_items, d_pool,
                     n, n_users, pool_size, n_neg, seed, d_out_items, d_out_labels, err_flag);
  RX_LAUNCH_CHECK();
  return 0;
}
