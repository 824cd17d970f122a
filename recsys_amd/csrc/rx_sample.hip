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

// ---------------------------------------------------------------- next batch of a resident epoch
// The train step of a MovieLens-sized batch takes tens of microseconds on the GPU; slicing the
// epoch tensors and copying the slices into the recorded step's static batch from the host costs
// more than that per step.  This kernel is recorded at the END of the step's graph: it adds the
// step's loss to the epoch's running sum and copies batch cursor + 1 of every epoch tensor into the
// static batch, so the host's share of a step is one graph launch.
// All blocks read the cursor on entry; the block that arrives last at the ticket (an integer
// atomic: no block can still be about to read) advances it and takes the loss.
__global__ void __launch_bounds__(RX_BLOCK)
epoch_advance_kernel(rx_batch_plan p, int64_t* __restrict__ cursor, uint32_t* __restrict__ ticket,
                     const void* __restrict__ loss, int loss_f64, double* __restrict__ running) {
  const int64_t next = cursor[0] + 1;
  const int64_t take = cursor[0] + (p.ahead > 0 ? p.ahead : 1);   // the batch that goes into the static tensors
  if (take >= 0 && take < p.n_batches) {
    for (int t = 0; t < p.count; ++t) {
      const int64_t words = p.rows_per_batch * (p.row_bytes[t] / 8);
      const uint64_t* src = reinterpret_cast<const uint64_t*>(p.src[t]) + take * words;
      uint64_t* dst = reinterpret_cast<uint64_t*>(p.dst[t]);
      for (int64_t i = (int64_t)blockIdx.x * RX_BLOCK + threadIdx.x; i < words; i += (int64_t)gridDim.x * RX_BLOCK) dst[i] = src[i];
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    const uint32_t arrived = atomicAdd(ticket, 1u);
    if (arrived == gridDim.x - 1) {
      ticket[0] = 0;
      cursor[0] = next;
      if (loss && running) running[0] += loss_f64 ? reinterpret_cast<const double*>(loss)[0] : (double)reinterpret_cast<const float*>(loss)[0];
    }
  }
}

extern "C" int rx_epoch_advance(const rx_batch_plan* plan, int64_t* d_cursor, uint32_t* d_ticket, const void* d_loss,
                                int loss_f64, double* d_running, int device, void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(plan && d_cursor && d_ticket, "rx_epoch_advance: null argument");
  RX_REQUIRE(plan->count >= 1 && plan->count <= RX_BATCH_MAX_TENSORS && plan->rows_per_batch >= 1 && plan->n_batches >= 0 && plan->ahead >= 0,
             "rx_epoch_advance: 1..%d tensors, a positive batch size and a non-negative batch count", RX_BATCH_MAX_TENSORS);
  RX_REQUIRE((d_loss == nullptr) == (d_running == nullptr), "rx_epoch_advance: loss and running sum come as a pair");
  int64_t most = 0;
  for (int t = 0; t < plan->count; ++t) {
    RX_REQUIRE(plan->src[t] && plan->dst[t] && plan->row_bytes[t] >= 8 && plan->row_bytes[t] % 8 == 0 &&
               ((uintptr_t)plan->src[t] & 7) == 0 && ((uintptr_t)plan->dst[t] & 7) == 0,
               "rx_epoch_advance: tensor %d needs 8-byte aligned buffers and rows of whole 8-byte words", t);
    const int64_t w = plan->rows_per_batch * (plan->row_bytes[t] / 8);
    most = w > most ? w : most;
  }
  int64_t g = rx_cdiv(most, RX_BLOCK * 4);
  const unsigned grid = (unsigned)(g < 1 ? 1 : (g > 128 ? 128 : g));
  hipLaunchKernelGGL(epoch_advance_kernel, dim3(grid), dim3(RX_BLOCK), 0, (hipStream_t)stream, *plan, d_cursor, d_ticket,
                     d_loss, loss_f64, d_running);
  RX_LAUNCH_CHECK();
  return 0;
}
