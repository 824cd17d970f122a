// Deterministic index preparation for the embedding backward (gfx950): a stable LSD radix
// sort of (row, position) pairs written for 64-lane wavefronts, then run-length compaction
// into distinct rows + run offsets.
//
// Sort structure per pass (3 launches; digits of up to 11 bits, spread evenly over the passes):
//   hist    each block counts the digits of its 4096-key tile (LDS integer atomics)
//   scan    every block scans 2048 of the [digit][block] counts in place and publishes its total
//   scatter first folds the scan of those totals into its prologue (they are few), then
//           each wave ranks its 1024 keys stably: per key, `bits` ballots give the set of lanes
//           with the same digit (a 64-lane "match"; digit width is a compile-time constant there,
//           all of a thread's keys are matched before any counter is touched), then key by key
//           every lane of a group reads the wave's private digit counter, the lowest lane advances
//           it, and the rank is counter + popcount(lanes below).
//           Wave counters are prefixed across the block's 4 waves, so the final order is
//           (block, wave, round, lane) = original order: the sort is stable by construction
//           and uses no float or global atomics.
#include "rx_common.h"

#define SORT_MAX_BITS 11
#define SORT_MAX_BINS (1 << SORT_MAX_BITS)
#define SORT_ITEMS 16
#define SORT_TILE (RX_BLOCK * SORT_ITEMS)  // keys per block
#define SORT_WAVES (RX_BLOCK / 64)
#define SORT_WAVE_KEYS (64 * SORT_ITEMS)
#define SCAN_TILE (RX_BLOCK * 8)           // entries of the count array per scan block

// The first pass reads the caller's int64 rows directly: key = row clamped into range (rows < 2^32),
// value = position.  Later passes read the uint32 pairs of the previous one.  RAW selects the source:
//   0  uint32 keys of the previous pass
//   1  a plain int64 list
//   2  the FM model's index matrices: position i < B*F is extra field i % F of sample i / F, shifted
//      by that field's base row; the rest are the B*N item rows shifted by item_base (the order
//      rx_fm2_flat_indices writes, without materialising that list)
struct RawSrc {
  const int64_t* idx;
  const int64_t* ex_idx;
  const int64_t* ex_base;
  const int64_t* item_idx;
  int64_t ex_ld, item_ld, item_base;
  uint32_t n_ex, F, N;
};

template <int RAW>
__device__ __forceinline__ uint32_t raw_key(const RawSrc& s, int64_t i, int64_t n_rows, bool* bad) {
  int64_t r;
  if constexpr (RAW == 2) {
    // branch-free, so that the loads of a thread's 16 keys are all in flight together
    const uint32_t u = (uint32_t)i;
    const bool is_ex = u < s.n_ex;
    const uint32_t q = is_ex ? u : u - s.n_ex;
    const uint32_t per = is_ex ? s.F : s.N;
    const uint32_t b = q / per, k = q - b * per;
    const int64_t* p = is_ex ? s.ex_idx + ((int64_t)b * s.ex_ld + k) : s.item_idx + ((int64_t)b * s.item_ld + k);
    const int64_t shift = s.ex_base[is_ex ? k : 0];
    r = *p + (is_ex ? shift : s.item_base);
  } else {
    r = s.idx[i];
  }
  *bad = r < 0 || r >= n_rows;
  return (uint32_t)(r < 0 ? 0 : (r >= n_rows ? n_rows - 1 : r));
}

// counts[digit][block] for this pass (bins = 1 << bits)
template <int RAW>
__global__ void __launch_bounds__(RX_BLOCK)
sort_hist_kernel(RawSrc src, const uint32_t* __restrict__ keys_in, int64_t n, int64_t n_rows, int shift, int bits,
                 uint32_t* __restrict__ block_hist, int n_blocks, int32_t* err_flag) {
  extern __shared__ uint32_t h[];
  const int bins = 1 << bits;
  for (int i = threadIdx.x; i < bins; i += blockDim.x) h[i] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * SORT_TILE;
  uint32_t key[SORT_ITEMS];
  bool any_bad = false;
#pragma unroll
  for (int it = 0; it < SORT_ITEMS; ++it) {
    const int64_t i = base + it * RX_BLOCK + threadIdx.x;
    const int64_t ii = i < n ? i : n - 1;
    if constexpr (RAW != 0) {
      bool bad;
      key[it] = raw_key<RAW>(src, ii, n_rows, &bad);
      any_bad |= bad;
    } else {
      key[it] = keys_in[ii];
    }
  }
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int it = 0; it < SORT_ITEMS; ++it) {
    const int64_t i = base + it * RX_BLOCK + threadIdx.x;
    const uint32_t d = (key[it] >> shift) & (bins - 1);
    bool todo = i < n;
    if constexpr (RAW == 0) {
      // High digits of row ids are skewed (a few digits hold most keys), and 64 lanes adding to one
      // LDS word serialise.  Two rounds of "everyone with the first pending lane's digit": that
      // lane adds the group's size once.  Whatever is left goes through plain atomics.
#pragma unroll
      for (int round = 0; round < 2; ++round) {
        const unsigned long long pend = __ballot(todo);
        if (pend == 0) break;
        const int leader = __ffsll((long long)pend) - 1;
        const uint32_t dl = (uint32_t)__shfl((int)d, leader, 64);
        const bool mine = todo && d == dl;
        const unsigned long long grp = __ballot(mine);
        if (lane == leader) atomicAdd(&h[dl], (uint32_t)__popcll(grp));
        todo = todo && !mine;
      }
    }
    if (todo) atomicAdd(&h[d], 1u);
  }
  if (RAW != 0 && any_bad && err_flag) atomicOr(err_flag, RX_ERR_INDEX);
  __syncthreads();
  for (int d = threadIdx.x; d < bins; d += blockDim.x) block_hist[(int64_t)d * n_blocks + blockIdx.x] = h[d];
}

// Exclusive scan of the digit-major count array in two levels: every block scans its own
// SCAN_TILE entries in place and publishes its total; one block then scans the totals; the
// scatter kernel adds the two.
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* lds4, uint32_t* total) {
  // wave inclusive scan by shuffles, then the 4 wave totals through LDS
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = v;
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(inc, off, 64);
    if (lane >= off) inc += t;
  }
  if (lane == 63) lds4[wave] = inc;
  __syncthreads();
  uint32_t base = 0;
  for (int w = 0; w < wave; ++w) base += lds4[w];
  if (total) *total = lds4[0] + lds4[1] + lds4[2] + lds4[3];
  __syncthreads();
  return base + inc - v;
}

__global__ void __launch_bounds__(RX_BLOCK)
scan_tiles_kernel(uint32_t* __restrict__ data, int64_t total, uint32_t* __restrict__ tile_sums) {
  __shared__ uint32_t lds4[4];
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * 8;
  uint32_t v[8], sum = 0;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    v[e] = (base + e < total) ? data[base + e] : 0u;
    sum += v[e];
  }
  uint32_t tile_total;
  uint32_t run = block_exclusive_scan(sum, lds4, &tile_total);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    if (base + e < total) data[base + e] = run;
    run += v[e];
  }
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = tile_total;
}

// In-place exclusive scan of up to a few hundred thousand tile totals by one block.
__global__ void __launch_bounds__(1024)
scan_totals_kernel(uint32_t* __restrict__ sums, int64_t count) {
  __shared__ uint32_t part[1024];
  const int t = threadIdx.x;
  const int64_t per = (count + 1023) / 1024;
  const int64_t lo = t * per, hi = (lo + per < count) ? lo + per : count;
  uint32_t s = 0;
  for (int64_t i = lo; i < hi; ++i) s += sums[i];
  part[t] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const uint32_t v = (t >= off) ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  uint32_t run = part[t] - s;
  for (int64_t i = lo; i < hi; ++i) {
    const uint32_t c = sums[i];
    sums[i] = run;
    run += c;
  }
}

// For each of a thread's keys: the set of live lanes of the wavefront whose digit equals this
// lane's (a 64-lane "match").  Per bit one ballot; sb is all ones where the lane's bit is set, so
// (ballot ^ sb) marks the lanes whose bit differs from mine.
template <int BITS>
__device__ __forceinline__ void match_all(const uint32_t (&key)[SORT_ITEMS], int shift, int64_t wbase, int lane,
                                          int64_t n, unsigned long long (&peers)[SORT_ITEMS]) {
#pragma unroll
  for (int it = 0; it < SORT_ITEMS; ++it) {
    const bool live = (wbase + it * 64 + lane) < n;
    const uint32_t d = (key[it] >> shift) & ((1u << BITS) - 1);
    unsigned long long p = __ballot(live);
#pragma unroll
    for (int bit = 0; bit < BITS; ++bit) {
      const int32_t sb = (int32_t)(d << (31 - bit)) >> 31;
      const unsigned long long m = __ballot(sb != 0);
      p &= ~(m ^ (unsigned long long)(long long)sb);
    }
    peers[it] = live ? p : 0ull;
  }
}

// `tile_sums` are the totals published by scan_tiles_kernel.  FOLD: they are still unscanned and
// at most SCAN_FOLD_MAX many, and every block scans them itself in LDS while its keys are in
// flight, which saves a single-block launch per pass.  Otherwise they arrive scanned.
#define SCAN_FOLD_MAX 1024

template <int RAW, bool FOLD>
__global__ void __launch_bounds__(RX_BLOCK)
sort_scatter_kernel(RawSrc src, const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, int64_t n,
                    int64_t n_rows, int shift, int bits, const uint32_t* __restrict__ block_off,
                    const uint32_t* __restrict__ tile_sums, int n_tiles, int n_blocks,
                    uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out) {
  // dynamic LDS: [bins] global start per digit | [bins] tile-local start per digit |
  //              max([SORT_WAVES][bins] wave counters, [2][SORT_TILE] staged (key, value) pairs)
  extern __shared__ uint32_t dyn[];
  __shared__ uint32_t tile_lds[FOLD ? SCAN_FOLD_MAX : 1];
  __shared__ uint32_t lds4[4];
  const int bins = 1 << bits;
  uint32_t* gstart = dyn;
  uint32_t* texcl = dyn + bins;
  uint32_t* wave_cnt = dyn + 2 * bins;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < SORT_WAVES * bins; i += blockDim.x) wave_cnt[i] = 0;

  uint32_t* my_cnt = wave_cnt + wave * bins;
  const int64_t tbase = (int64_t)blockIdx.x * SORT_TILE;
  const int64_t wbase = tbase + (int64_t)wave * SORT_WAVE_KEYS;
  uint32_t key[SORT_ITEMS], val[SORT_ITEMS], rank[SORT_ITEMS];
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int it = 0; it < SORT_ITEMS; ++it) {
    const int64_t i = wbase + it * 64 + lane;
    const bool live = i < n;
    const int64_t ii = live ? i : n - 1;
    if constexpr (RAW != 0) {
      bool bad;
      key[it] = raw_key<RAW>(src, ii, n_rows, &bad);
      val[it] = (uint32_t)ii;
    } else {
      key[it] = keys_in[ii];
      val[it] = vals_in[ii];
    }
    if (!live) key[it] = 0xFFFFFFFFu;
  }
  if constexpr (FOLD) {
    // exclusive scan of the tile totals: four per thread, then the block scan
    uint32_t v[SCAN_FOLD_MAX / RX_BLOCK], sum = 0;
#pragma unroll
    for (int e = 0; e < SCAN_FOLD_MAX / RX_BLOCK; ++e) {
      const int t = threadIdx.x * (SCAN_FOLD_MAX / RX_BLOCK) + e;
      v[e] = t < n_tiles ? tile_sums[t] : 0u;
      sum += v[e];
    }
    const uint32_t run0 = block_exclusive_scan(sum, lds4, nullptr);
    uint32_t run = run0;
#pragma unroll
    for (int e = 0; e < SCAN_FOLD_MAX / RX_BLOCK; ++e) {
      tile_lds[threadIdx.x * (SCAN_FOLD_MAX / RX_BLOCK) + e] = run;
      run += v[e];
    }
  }
  __syncthreads();
  // lanes (live ones) holding the same digit, for all of the thread's keys first: the ballot chains
  // of different keys are independent and interleave
  unsigned long long peers[SORT_ITEMS];
  switch (bits) {  // wave-uniform; the digit width as a compile-time constant unrolls the match
#define MATCH_CASE(B) case B: match_all<B>(key, shift, wbase, lane, n, peers); break;
    MATCH_CASE(1) MATCH_CASE(2) MATCH_CASE(3) MATCH_CASE(4) MATCH_CASE(5) MATCH_CASE(6)
    MATCH_CASE(7) MATCH_CASE(8) MATCH_CASE(9) MATCH_CASE(10) MATCH_CASE(11)
#undef MATCH_CASE
  }
  // then the counters, key by key: every lane of a group reads the digit's counter (one LDS
  // broadcast), the group's lowest lane advances it.  LDS operations of a wavefront execute in
  // program order, so the next key's read sees this write.
#pragma unroll
  for (int it = 0; it < SORT_ITEMS; ++it) {
    const uint32_t d = (key[it] >> shift) & (bins - 1);
    const unsigned long long p = peers[it];
    uint32_t base = 0;
    if (p) base = my_cnt[d];
    if (p && lane == __ffsll((long long)p) - 1) my_cnt[d] = base + (uint32_t)__popcll(p);
    rank[it] = base + (uint32_t)__popcll(p & lt_mask);
  }
  __syncthreads();
  // per digit: exclusive prefix over the block's waves (tile-local), the tile's count, and the
  // digit's global start for this tile
  for (int d = threadIdx.x; d < bins; d += blockDim.x) {
    const int64_t e = (int64_t)d * n_blocks + blockIdx.x;
    gstart[d] = block_off[e] + (FOLD ? tile_lds[e / SCAN_TILE] : tile_sums[e / SCAN_TILE]);
    uint32_t run = 0;
#pragma unroll
    for (int w = 0; w < SORT_WAVES; ++w) {
      const uint32_t c = wave_cnt[w * bins + d];
      wave_cnt[w * bins + d] = run;
      run += c;
    }
    texcl[d] = run;
  }
  __syncthreads();
  {
    // exclusive scan of the tile's digit counts (bins / 256 consecutive digits per thread)
    const int per = (bins + RX_BLOCK - 1) / RX_BLOCK;
    uint32_t v[SORT_MAX_BINS / RX_BLOCK], sum = 0;
#pragma unroll
    for (int e = 0; e < SORT_MAX_BINS / RX_BLOCK; ++e) {
      const int d = threadIdx.x * per + e;
      v[e] = (e < per && d < bins) ? texcl[d] : 0u;
      sum += v[e];
    }
    uint32_t run = block_exclusive_scan(sum, lds4, nullptr);
#pragma unroll
    for (int e = 0; e < SORT_MAX_BINS / RX_BLOCK; ++e) {
      const int d = threadIdx.x * per + e;
      if (e < per && d < bins) texcl[d] = run;
      run += v[e];
    }
  }
  __syncthreads();
  // position of every key inside the sorted tile
#pragma unroll
  for (int it = 0; it < SORT_ITEMS; ++it) {
    const uint32_t d = (key[it] >> shift) & (bins - 1);
    rank[it] += texcl[d] + my_cnt[d];
  }
  __syncthreads();
  // The wave counters are dead: their space stages the tile in sorted order, so that the
  // write-out below sends neighbouring lanes to neighbouring addresses (runs of equal digits are
  // contiguous in the output) instead of one 4-byte store per key and destination.
  uint32_t* sk = wave_cnt;
  uint32_t* sv = wave_cnt + SORT_TILE;
#pragma unroll
  for (int it = 0; it < SORT_ITEMS; ++it) {
    if (wbase + it * 64 + lane < n) {
      sk[rank[it]] = key[it];
      sv[rank[it]] = val[it];
    }
  }
  __syncthreads();
  const int tile_n = (int)((tbase + SORT_TILE <= n) ? SORT_TILE : n - tbase);
#pragma unroll
  for (int it = 0; it < SORT_ITEMS; ++it) {
    const int j = it * RX_BLOCK + threadIdx.x;
    if (j < tile_n) {
      const uint32_t k = sk[j];
      const uint32_t d = (k >> shift) & (bins - 1);
      const uint32_t dst = gstart[d] + ((uint32_t)j - texcl[d]);
      keys_out[dst] = k;
      vals_out[dst] = sv[j];
    }
  }
}

// ---------------------------------------------------------------- run-length compaction
#define DEDUP_ITEMS 8
#define DEDUP_TILE (RX_BLOCK * DEDUP_ITEMS)

// a thread's DEDUP_ITEMS consecutive keys (vector loads when the whole group is inside the list),
// and which of them start a run
__device__ __forceinline__ uint32_t dedup_heads(const uint32_t* __restrict__ keys, int64_t base, int64_t n,
                                                uint32_t (&k)[DEDUP_ITEMS], bool (&head)[DEDUP_ITEMS]) {
  if (base + DEDUP_ITEMS <= n) {
#pragma unroll
    for (int q = 0; q < DEDUP_ITEMS / 4; ++q) {
      const uint4 v = *(const uint4*)(keys + base + 4 * q);
      k[4 * q] = v.x; k[4 * q + 1] = v.y; k[4 * q + 2] = v.z; k[4 * q + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int it = 0; it < DEDUP_ITEMS; ++it) k[it] = base + it < n ? keys[base + it] : 0u;
  }
  const uint32_t prev = (base > 0 && base < n) ? keys[base - 1] : 0u;
  uint32_t c = 0;
#pragma unroll
  for (int it = 0; it < DEDUP_ITEMS; ++it) {
    const uint32_t before = it == 0 ? prev : k[it - 1];
    head[it] = base + it < n && (base + it == 0 || k[it] != before);
    c += head[it] ? 1u : 0u;
  }
  return c;
}

__global__ void __launch_bounds__(RX_BLOCK)
dedup_count_kernel(const uint32_t* __restrict__ keys, int64_t n, uint32_t* __restrict__ block_counts) {
  __shared__ uint32_t wsum[RX_BLOCK / 64];
  const int64_t base = (int64_t)blockIdx.x * DEDUP_TILE + (int64_t)threadIdx.x * DEDUP_ITEMS;
  uint32_t k[DEDUP_ITEMS];
  bool head[DEDUP_ITEMS];
  uint32_t c = dedup_heads(keys, base, n, k, head);
  for (int m = 32; m > 0; m >>= 1) c += __shfl_xor(c, m, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// FOLD: block_counts holds the raw per-block head counts (at most SCAN_FOLD_MAX of them) and every
// block sums the ones before it; otherwise their exclusive scan.
template <bool FOLD>
__global__ void __launch_bounds__(RX_BLOCK)
dedup_write_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals, int64_t n,
                   const uint32_t* __restrict__ block_base, int n_blocks,
                   int32_t* __restrict__ sorted_pos, int64_t* __restrict__ uniq_rows,
                   int32_t* __restrict__ seg_off, int32_t* __restrict__ n_uniq,
                   int32_t* __restrict__ inverse, int32_t* __restrict__ sorted_slot) {
  __shared__ uint32_t lds4[4];
  __shared__ uint32_t before_lds[4];
  const int64_t base = (int64_t)blockIdx.x * DEDUP_TILE + (int64_t)threadIdx.x * DEDUP_ITEMS;
  bool head[DEDUP_ITEMS];
  uint32_t k[DEDUP_ITEMS], v[DEDUP_ITEMS];
  const uint32_t c = dedup_heads(keys, base, n, k, head);
  if (base + DEDUP_ITEMS <= n) {
#pragma unroll
    for (int q = 0; q < DEDUP_ITEMS / 4; ++q) {
      const uint4 t = *(const uint4*)(vals + base + 4 * q);
      v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
    }
  } else {
#pragma unroll
    for (int it = 0; it < DEDUP_ITEMS; ++it) v[it] = base + it < n ? vals[base + it] : 0u;
  }
  uint32_t first;
  if constexpr (FOLD) {
    // heads in the blocks before this one: SCAN_FOLD_MAX / RX_BLOCK counts per thread, fixed tree
    uint32_t mine = 0;
#pragma unroll
    for (int e = 0; e < SCAN_FOLD_MAX / RX_BLOCK; ++e) {
      const int b = e * RX_BLOCK + threadIdx.x;
      if (b < (int)blockIdx.x) mine += block_base[b];
    }
    for (int m = 32; m > 0; m >>= 1) mine += __shfl_xor(mine, m, 64);
    if ((threadIdx.x & 63) == 0) before_lds[threadIdx.x >> 6] = mine;
    __syncthreads();
    first = before_lds[0] + before_lds[1] + before_lds[2] + before_lds[3];
  } else {
    first = block_base[blockIdx.x];
  }
  uint32_t tile_total;
  uint32_t slot = first + block_exclusive_scan(c, lds4, &tile_total);
  int32_t slots[DEDUP_ITEMS];
#pragma unroll
  for (int it = 0; it < DEDUP_ITEMS; ++it) {
    const int64_t i = base + it;
    if (head[it]) {
      uniq_rows[slot] = (int64_t)k[it];
      seg_off[slot] = (int32_t)i;
      ++slot;
    }
    slots[it] = (int32_t)(slot - 1);
    if (i < n && inverse) inverse[v[it]] = slots[it];
  }
  const bool out16 = (((uintptr_t)sorted_pos | (uintptr_t)sorted_slot) & 15) == 0;
  if (out16 && base + DEDUP_ITEMS <= n) {
#pragma unroll
    for (int q = 0; q < DEDUP_ITEMS / 4; ++q) {
      *(int4*)(sorted_pos + base + 4 * q) = make_int4((int)v[4 * q], (int)v[4 * q + 1], (int)v[4 * q + 2], (int)v[4 * q + 3]);
      if (sorted_slot) *(int4*)(sorted_slot + base + 4 * q) = make_int4(slots[4 * q], slots[4 * q + 1], slots[4 * q + 2], slots[4 * q + 3]);
    }
  } else {
#pragma unroll
    for (int it = 0; it < DEDUP_ITEMS; ++it)
      if (base + it < n) {
        sorted_pos[base + it] = (int32_t)v[it];
        if (sorted_slot) sorted_slot[base + it] = slots[it];
      }
  }
  if (blockIdx.x == n_blocks - 1 && threadIdx.x == 0) {
    const uint32_t total = first + tile_total;
    n_uniq[0] = (int32_t)total;
    seg_off[total] = (int32_t)n;
  }
}

__global__ void __launch_bounds__(RX_BLOCK)
add_tile_offsets_kernel(uint32_t* __restrict__ data, int64_t total, const uint32_t* __restrict__ tile_off) {
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * 8;
  const uint32_t off = tile_off[blockIdx.x];
#pragma unroll
  for (int e = 0; e < 8; ++e)
    if (base + e < total) data[base + e] += off;
}

// ---------------------------------------------------------------- short lists: one launch
// A list of at most SMALL_MAX positions (the user / item lists of a MovieLens-sized batch) is
// sorted and compacted by ONE workgroup: 16 wavefronts, 8 keys per lane, the pairs staged in LDS
// between the passes.  The ranking is the one of sort_scatter_kernel (ballot match inside the
// wavefront, wavefront-private digit counters, prefix over the wavefronts), so the order is the
// same stable order and every output equals that of the multi-launch path; what goes away is a
// chain of about ten dependent launches that cost more than the work.
#define SMALL_THREADS 1024
#define SMALL_ITEMS 8
#define SMALL_MAX (SMALL_THREADS * SMALL_ITEMS)
#define SMALL_WAVES (SMALL_THREADS / 64)
#define SMALL_WAVE_KEYS (64 * SMALL_ITEMS)
#define SMALL_BITS 8
#define SMALL_BINS (1 << SMALL_BITS)

__device__ __forceinline__ uint32_t small_block_scan(uint32_t v, uint32_t* lds16, uint32_t* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = v;
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(inc, off, 64);
    if (lane >= off) inc += t;
  }
  if (lane == 63) lds16[wave] = inc;
  __syncthreads();
  uint32_t base = 0, all = 0;
#pragma unroll
  for (int w = 0; w < SMALL_WAVES; ++w) {
    const uint32_t c = lds16[w];
    base += w < wave ? c : 0u;
    all += c;
  }
  if (total) *total = all;
  __syncthreads();
  return base + inc - v;
}

template <int BITS>
__device__ __forceinline__ void small_match(const uint32_t (&key)[SMALL_ITEMS], int shift, int wbase, int lane, int n,
                                            unsigned long long (&peers)[SMALL_ITEMS]) {
#pragma unroll
  for (int it = 0; it < SMALL_ITEMS; ++it) {
    const bool live = (wbase + it * 64 + lane) < n;
    const uint32_t d = (key[it] >> shift) & ((1u << BITS) - 1);
    unsigned long long p = __ballot(live);
#pragma unroll
    for (int bit = 0; bit < BITS; ++bit) {
      const int32_t sb = (int32_t)(d << (31 - bit)) >> 31;
      const unsigned long long m = __ballot(sb != 0);
      p &= ~(m ^ (unsigned long long)(long long)sb);
    }
    peers[it] = live ? p : 0ull;
  }
}

template <int RAW>
__global__ void __launch_bounds__(SMALL_THREADS)
sort_dedup_small_kernel(RawSrc src, int n, int64_t n_rows, int passes, int bits, int32_t* __restrict__ sorted_pos,
                        int32_t* __restrict__ sorted_slot, int64_t* __restrict__ uniq_rows,
                        int32_t* __restrict__ seg_off, int32_t* __restrict__ n_uniq, int32_t* __restrict__ inverse,
                        int32_t* err_flag) {
  // dynamic LDS: [SMALL_MAX] keys | [SMALL_MAX] positions | [SMALL_WAVES][SMALL_BINS] counters | [SMALL_BINS] digit starts
  extern __shared__ uint32_t dyn[];
  __shared__ uint32_t lds16[SMALL_WAVES];
  uint32_t* sk = dyn;
  uint32_t* sv = dyn + SMALL_MAX;
  uint32_t* wave_cnt = dyn + 2 * SMALL_MAX;
  uint32_t* dstart = wave_cnt + SMALL_WAVES * SMALL_BINS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wbase = wave * SMALL_WAVE_KEYS;
  const int bins = 1 << bits;
  uint32_t* my_cnt = wave_cnt + wave * bins;
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  uint32_t key[SMALL_ITEMS], val[SMALL_ITEMS], rank[SMALL_ITEMS];
  bool any_bad = false;
#pragma unroll
  for (int it = 0; it < SMALL_ITEMS; ++it) {
    const int i = wbase + it * 64 + lane;
    const int ii = i < n ? i : n - 1;
    bool bad;
    key[it] = raw_key<RAW>(src, ii, n_rows, &bad);
    any_bad |= bad;
    val[it] = (uint32_t)ii;
  }
  if (any_bad && err_flag) atomicOr(err_flag, RX_ERR_INDEX);
  for (int p = 0; p < passes; ++p) {
    const int shift = p * bits;
    for (int i = threadIdx.x; i < SMALL_WAVES * bins; i += SMALL_THREADS) wave_cnt[i] = 0;
    __syncthreads();
    unsigned long long peers[SMALL_ITEMS];
    switch (bits) {
#define SMALL_CASE(B) case B: small_match<B>(key, shift, wbase, lane, n, peers); break;
      SMALL_CASE(1) SMALL_CASE(2) SMALL_CASE(3) SMALL_CASE(4) SMALL_CASE(5) SMALL_CASE(6) SMALL_CASE(7) SMALL_CASE(8)
#undef SMALL_CASE
    }
#pragma unroll
    for (int it = 0; it < SMALL_ITEMS; ++it) {
      const uint32_t d = (key[it] >> shift) & (bins - 1);
      const unsigned long long pe = peers[it];
      uint32_t base = 0;
      if (pe) base = my_cnt[d];
      if (pe && lane == __ffsll((long long)pe) - 1) my_cnt[d] = base + (uint32_t)__popcll(pe);
      rank[it] = base + (uint32_t)__popcll(pe & lt_mask);
    }
    __syncthreads();
    // per digit: exclusive prefix over the wavefronts, and the digit's total
    uint32_t total_d = 0;
    if ((int)threadIdx.x < bins) {
      const int d = threadIdx.x;
#pragma unroll
      for (int w = 0; w < SMALL_WAVES; ++w) {
        const uint32_t c = wave_cnt[w * bins + d];
        wave_cnt[w * bins + d] = total_d;
        total_d += c;
      }
    }
    const uint32_t start_d = small_block_scan(total_d, lds16, nullptr);   // (syncs inside)
    if ((int)threadIdx.x < bins) dstart[threadIdx.x] = start_d;
    __syncthreads();
#pragma unroll
    for (int it = 0; it < SMALL_ITEMS; ++it) {
      const uint32_t d = (key[it] >> shift) & (bins - 1);
      if (wbase + it * 64 + lane < n) {
        const uint32_t r = rank[it] + dstart[d] + my_cnt[d];
        sk[r] = key[it];
        sv[r] = val[it];
      }
    }
    __syncthreads();
    if (p + 1 < passes) {
#pragma unroll
      for (int it = 0; it < SMALL_ITEMS; ++it) {
        const int i = wbase + it * 64 + lane;
        const int ii = i < n ? i : n - 1;
        key[it] = sk[ii];
        val[it] = sv[ii];
      }
      __syncthreads();
    }
  }
  // run-length compaction: a thread takes SMALL_ITEMS consecutive sorted entries
  const int base = threadIdx.x * SMALL_ITEMS;
  uint32_t k[SMALL_ITEMS], v[SMALL_ITEMS];
  bool head[SMALL_ITEMS];
  uint32_t c = 0;
  const uint32_t prev = (base > 0 && base < n) ? sk[base - 1] : 0u;
#pragma unroll
  for (int it = 0; it < SMALL_ITEMS; ++it) {
    const int i = base + it;
    k[it] = i < n ? sk[i] : 0u;
    v[it] = i < n ? sv[i] : 0u;
    const uint32_t before = it == 0 ? prev : k[it - 1];
    head[it] = i < n && (i == 0 || k[it] != before);
    c += head[it] ? 1u : 0u;
  }
  uint32_t total;
  uint32_t slot = small_block_scan(c, lds16, &total);
#pragma unroll
  for (int it = 0; it < SMALL_ITEMS; ++it) {
    const int i = base + it;
    if (head[it]) {
      uniq_rows[slot] = (int64_t)k[it];
      seg_off[slot] = i;
      ++slot;
    }
    if (i < n) {
      sorted_pos[i] = (int32_t)v[it];
      if (sorted_slot) sorted_slot[i] = (int32_t)(slot - 1);
      if (inverse) inverse[v[it]] = (int32_t)(slot - 1);
    }
  }
  if (threadIdx.x == 0) {
    n_uniq[0] = (int32_t)total;
    seg_off[total] = n;
  }
}

static int sort_dedup_small(const RawSrc& raw, int raw_mode, int64_t n, int64_t n_rows, int32_t* d_sorted_pos,
                            int32_t* d_sorted_slot, int64_t* d_uniq_rows, int32_t* d_seg_off, int32_t* d_n_uniq,
                            int32_t* d_inverse, int32_t* err_flag, hipStream_t st) {
  int bits = 1;
  while (bits < 32 && (1ull << bits) < (unsigned long long)n_rows) ++bits;
  const int passes = (bits + SMALL_BITS - 1) / SMALL_BITS;
  const int per = (bits + passes - 1) / passes;
  const size_t lds = (2 * (size_t)SMALL_MAX + SMALL_WAVES * SMALL_BINS + SMALL_BINS) * sizeof(uint32_t);
  // LDS beyond 64 KiB per block has to be asked for, per kernel and device
  static bool attr_set[2][64] = {};
  int device = 0;
  RX_HIP(hipGetDevice(&device));
  const bool known = device >= 0 && device < 64;
  if (raw_mode == 2) {
    if (!known || !attr_set[1][device]) {
      RX_HIP(hipFuncSetAttribute((const void*)sort_dedup_small_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      if (known) attr_set[1][device] = true;
    }
    hipLaunchKernelGGL(sort_dedup_small_kernel<2>, dim3(1), dim3(SMALL_THREADS), lds, st, raw, (int)n, n_rows, passes, per,
                       d_sorted_pos, d_sorted_slot, d_uniq_rows, d_seg_off, d_n_uniq, d_inverse, err_flag);
  } else {
    if (!known || !attr_set[0][device]) {
      RX_HIP(hipFuncSetAttribute((const void*)sort_dedup_small_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      if (known) attr_set[0][device] = true;
    }
    hipLaunchKernelGGL(sort_dedup_small_kernel<1>, dim3(1), dim3(SMALL_THREADS), lds, st, raw, (int)n, n_rows, passes, per,
                       d_sorted_pos, d_sorted_slot, d_uniq_rows, d_seg_off, d_n_uniq, d_inverse, err_flag);
  }
  RX_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- few distinct rows: one launch, a workgroup per row range
// When the table is small (the user and item tables of a MovieLens-sized model) the sort by row is a grouping
// into few buckets, and the workgroups can split the ROWS instead of the positions: workgroup j owns the rows
// [64 j, 64 j + 64), reads the whole list, and places the positions of its rows.  Everything it needs from the
// rest of the list -- how many positions hold a lower row, how many distinct lower rows occur -- it counts
// itself on the way, so no workgroup waits for another and one launch does what the general path does in
// eight.  Order inside a row is position order: every wavefront walks a contiguous share of the list once and
// queues the positions of the range (at most 65536 positions in all: the queues live in LDS); it then ranks the
// queued positions, 64 a step, by a 6-ballot match on the row, keeps wavefront-private counters, and the counters
// are prefixed over the wavefronts; a last walk over the queue repeats the ranking and writes.  Outputs are those of the other
// two paths, bit for bit.
#define KR_THREADS 1024
#define KR_WAVES (KR_THREADS / 64)
#define KR_KEYS 64
#define KR_MAX_ROWS 16384
#define KR_MAX_READS (4ll << 20)      // positions read over all workgroups: beyond this the general path is cheaper

__device__ __forceinline__ unsigned long long kr_match(bool in, uint32_t d) {
  unsigned long long p = __ballot(in);
#pragma unroll
  for (int bit = 0; bit < 6; ++bit) {
    const bool b = (d >> bit) & 1u;
    const unsigned long long m = __ballot(b);
    p &= b ? m : ~m;
  }
  return in ? p : 0ull;
}

#define KR_MAX_N 65536

// Dynamic LDS: per wavefront the queue of its positions whose row lies in the range (16-bit offsets into the
// wavefront's share, in position order: at worst the whole share), and one byte per row below the range (plain
// stores: which lower rows occur).  The list itself is read once; the few queued positions are read again (two
// more times) for their rows, which costs less than a place to keep them.
template <int RAW>
__global__ void __launch_bounds__(KR_THREADS)
sort_dedup_keyrange_kernel(RawSrc src, int n, int64_t n_rows, int32_t* __restrict__ sorted_pos, int32_t* __restrict__ sorted_slot,
                           int64_t* __restrict__ uniq_rows, int32_t* __restrict__ seg_off, int32_t* __restrict__ n_uniq,
                           int32_t* __restrict__ inverse, int32_t* err_flag) {
  extern __shared__ uint32_t kr_dyn[];
  __shared__ uint32_t cnt[KR_WAVES][KR_KEYS];    // positions of each row per wavefront; then the exclusive prefix over the wavefronts
  __shared__ uint32_t run[KR_WAVES][KR_KEYS];    // the last walk's running counts
  __shared__ uint32_t below_w[KR_WAVES], dist_w[KR_WAVES];
  __shared__ uint32_t kstart[KR_KEYS], kslot[KR_KEYS];
  __shared__ uint32_t below_all;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t lo = blockIdx.x * KR_KEYS;
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const int per_wave = (((n + KR_WAVES - 1) / KR_WAVES + 63) >> 6) << 6;
  uint16_t* queue = reinterpret_cast<uint16_t*>(kr_dyn) + wave * per_wave;
  uint32_t* seen_w = kr_dyn + ((per_wave * KR_WAVES) >> 1);
  uint8_t* seen = reinterpret_cast<uint8_t*>(seen_w);
  const int seen_words = (int)(lo >> 2);
  for (int i = tid; i < KR_WAVES * KR_KEYS; i += KR_THREADS) { (&cnt[0][0])[i] = 0; (&run[0][0])[i] = 0; }
  for (int i = tid; i < seen_words; i += KR_THREADS) seen_w[i] = 0;
  __syncthreads();
  const int w0 = wave * per_wave;
  const int steps = per_wave >> 6;
  uint32_t below = 0, qn = 0;      // (wavefront-uniform)
  bool any_bad = false;
  for (int s0 = 0; s0 < steps; s0 += 16) {       // sixteen loads in flight, then their sixteen steps
    uint32_t kk[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int i = w0 + (s0 + e) * 64 + lane;
      bool bad;
      kk[e] = raw_key<RAW>(src, i < n ? i : n - 1, n_rows, &bad);
      any_bad |= i < n && bad;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int s = s0 + e;
      const bool live = s < steps && w0 + s * 64 + lane < n;
      const uint32_t k = kk[e];
      const bool under = live && k < lo;
      if (under) seen[k] = 1;
      below += (uint32_t)__popcll(__ballot(under));
      const bool in = live && (k - lo) < KR_KEYS;
      const unsigned long long b = __ballot(in);
      if (in) queue[qn + (uint32_t)__popcll(b & lt_mask)] = (uint16_t)(s * 64 + lane);
      qn += (uint32_t)__popcll(b);
    }
  }
  if (any_bad && err_flag) atomicOr(err_flag, RX_ERR_INDEX);
  if (lane == 0) below_w[wave] = below;
  for (uint32_t e0 = 0; e0 < qn; e0 += 64) {
    const uint32_t e = e0 + lane;
    const bool has = e < qn;
    bool bad;
    const uint32_t k = raw_key<RAW>(src, w0 + (int)queue[has ? e : 0], n_rows, &bad);
    const uint32_t d = k - lo;
    const unsigned long long peers = kr_match(has, d);
    if (has && lane == __ffsll((long long)peers) - 1) cnt[wave][d] += (uint32_t)__popcll(peers);
  }
  __syncthreads();
  {
    uint32_t dist = 0;
    for (int i = tid; i < seen_words; i += KR_THREADS) dist += (uint32_t)__popc(seen_w[i] & 0x01010101u);
    for (int m = 32; m > 0; m >>= 1) dist += __shfl_xor(dist, m, 64);
    if (lane == 0) dist_w[wave] = dist;
  }
  __syncthreads();
  if (tid < 64) {
    uint32_t total = 0;
#pragma unroll
    for (int w = 0; w < KR_WAVES; ++w) {
      const uint32_t c = cnt[w][tid];
      cnt[w][tid] = total;
      total += c;
    }
    uint32_t inc = total;
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = __shfl_up(inc, off, 64);
      if (lane >= off) inc += t;
    }
    uint32_t ball = 0, dist = 0;
#pragma unroll
    for (int w = 0; w < KR_WAVES; ++w) { ball += below_w[w]; dist += dist_w[w]; }
    const bool present = total > 0;
    const unsigned long long pm = __ballot(present);
    const uint32_t slot = dist + (uint32_t)__popcll(pm & lt_mask);
    kstart[tid] = inc - total;
    kslot[tid] = slot;
    if (present) {
      uniq_rows[slot] = (int64_t)(lo + tid);
      seg_off[slot] = (int32_t)(ball + inc - total);
    }
    if (tid == 0) {
      below_all = ball;
      if (blockIdx.x == gridDim.x - 1) {
        const uint32_t nu = dist + (uint32_t)__popcll(pm);
        n_uniq[0] = (int32_t)nu;
        seg_off[nu] = n;
      }
    }
  }
  __syncthreads();
  const uint32_t ball = below_all;
  for (uint32_t e0 = 0; e0 < qn; e0 += 64) {
    const uint32_t e = e0 + lane;
    const bool has = e < qn;
    const int i = w0 + (int)queue[has ? e : 0];
    bool bad;
    const uint32_t k = raw_key<RAW>(src, i, n_rows, &bad);
    const uint32_t d = k - lo;
    const unsigned long long peers = kr_match(has, d);
    uint32_t base = 0;
    if (has) base = run[wave][d];
    if (has && lane == __ffsll((long long)peers) - 1) run[wave][d] = base + (uint32_t)__popcll(peers);
    if (has) {
      const uint32_t r = ball + kstart[d] + cnt[wave][d] + base + (uint32_t)__popcll(peers & lt_mask);
      const int32_t slot = (int32_t)kslot[d];
      sorted_pos[r] = i;
      if (sorted_slot) sorted_slot[r] = slot;
      if (inverse) inverse[i] = slot;
    }
  }
}

static size_t sort_keyrange_lds(int64_t n, int64_t n_rows) {
  const int64_t per_wave = (((n + KR_WAVES - 1) / KR_WAVES + 63) >> 6) << 6;
  return (size_t)(per_wave * KR_WAVES * 2 + rx_cdiv(n_rows, KR_KEYS) * KR_KEYS + 16);
}

static bool sort_keyrange_fits(int raw_mode, int64_t n, int64_t n_rows) {
  if (raw_mode != 1 || n < 1 || n > KR_MAX_N || n_rows > KR_MAX_ROWS) return false;
  return rx_cdiv(n_rows, KR_KEYS) * n <= KR_MAX_READS;
}

// Digit width: the fewest passes with digits of at most SORT_MAX_BITS bits, spread evenly.
static void sort_plan(int64_t n_rows, int* passes, int* bits_per_pass) {
  int bits = 1;
  while (bits < 32 && (1ull << bits) < (unsigned long long)n_rows) ++bits;
  *passes = (bits + SORT_MAX_BITS - 1) / SORT_MAX_BITS;
  *bits_per_pass = (bits + *passes - 1) / *passes;
}

// workspace layout: keys A | keys B | vals A | vals B | block hist | tile sums | dedup block counts
struct SortWs {
  uint32_t *ka, *kb, *va, *vb, *hist, *tiles, *counts;
  size_t bytes;
};

static SortWs sort_ws_layout(void* ws, int64_t n) {
  SortWs w;
  const int64_t nn = n > 0 ? n : 1;
  const size_t n_al = rx_align_up((size_t)nn * 4, 256);
  const int64_t nb = rx_cdiv(nn, SORT_TILE);
  const int64_t hist_entries = nb * SORT_MAX_BINS;
  const int64_t ndb = rx_cdiv(nn, DEDUP_TILE);
  char* p = (char*)ws;
  w.ka = (uint32_t*)p; p += n_al;
  w.kb = (uint32_t*)p; p += n_al;
  w.va = (uint32_t*)p; p += n_al;
  w.vb = (uint32_t*)p; p += n_al;
  w.hist = (uint32_t*)p; p += rx_align_up((size_t)hist_entries * 4, 256);
  w.tiles = (uint32_t*)p; p += rx_align_up((size_t)(rx_cdiv(hist_entries, SCAN_TILE) + rx_cdiv(ndb, SCAN_TILE) + 2) * 4, 256);
  w.counts = (uint32_t*)p; p += rx_align_up((size_t)ndb * 4, 256);
  w.bytes = (size_t)(p - (char*)ws);
  return w;
}

extern "C" size_t rx_sort_dedup_workspace_bytes(int64_t n) { return sort_ws_layout(nullptr, n).bytes; }

// one stable counting pass on `bits` bits starting at `shift`.  raw_mode != 0: first pass of the
// row sort, keys taken from `raw` (see RawSrc).  fold: the scatter scans the tile totals itself;
// callers that read w.tiles afterwards ask for the separately scanned form.
static int sort_pass(const SortWs& w, const RawSrc& raw, int raw_mode, int64_t n_rows, int32_t* err_flag,
                     const uint32_t* kin, const uint32_t* vin, uint32_t* kout, uint32_t* vout, int64_t n, int shift,
                     int bits, bool allow_fold, hipStream_t st) {
  const int nb = (int)rx_cdiv(n, SORT_TILE);
  const int bins = 1 << bits;
  const int64_t entries = (int64_t)nb * bins;
  const int n_tiles = (int)rx_cdiv(entries, SCAN_TILE);
  const bool fold = allow_fold && n_tiles <= SCAN_FOLD_MAX;
  const size_t hl = bins * sizeof(uint32_t);
  const size_t stage = (size_t)(SORT_WAVES * bins > 2 * SORT_TILE ? SORT_WAVES * bins : 2 * SORT_TILE);
  const size_t sl = (2 * (size_t)bins + stage) * sizeof(uint32_t);
#define HIST(RAWV) \
  hipLaunchKernelGGL(sort_hist_kernel<RAWV>, dim3(nb), dim3(RX_BLOCK), hl, st, raw, kin, n, n_rows, shift, bits, w.hist, nb, err_flag)
  if (raw_mode == 2) HIST(2);
  else if (raw_mode == 1) HIST(1);
  else HIST(0);
#undef HIST
  RX_LAUNCH_CHECK();
  hipLaunchKernelGGL(scan_tiles_kernel, dim3(n_tiles), dim3(RX_BLOCK), 0, st, w.hist, entries, w.tiles);
  RX_LAUNCH_CHECK();
  if (!fold) {
    hipLaunchKernelGGL(scan_totals_kernel, dim3(1), dim3(1024), 0, st, w.tiles, (int64_t)n_tiles);
    RX_LAUNCH_CHECK();
  }
#define SCATTER(RAWV, FOLDV) \
  hipLaunchKernelGGL((sort_scatter_kernel<RAWV, FOLDV>), dim3(nb), dim3(RX_BLOCK), sl, st, raw, kin, vin, n, n_rows, shift, bits, w.hist, w.tiles, n_tiles, nb, kout, vout)
  if (raw_mode == 2) { if (fold) SCATTER(2, true); else SCATTER(2, false); }
  else if (raw_mode == 1) { if (fold) SCATTER(1, true); else SCATTER(1, false); }
  else { if (fold) SCATTER(0, true); else SCATTER(0, false); }
#undef SCATTER
  RX_LAUNCH_CHECK();
  return 0;
}

// RX_SORT_GENERAL=1 in the environment keeps short lists on the multi-launch path (A/B timing, and the
// test that compares the two paths)
static const bool rx_sort_force_general = [] { const char* e = getenv("RX_SORT_GENERAL"); return e && e[0] == '1'; }();

// RX_SORT_NO_KEYRANGE=1 keeps small tables off the workgroup-per-row-range path (A/B timing)
static const bool rx_sort_no_keyrange = [] { const char* e = getenv("RX_SORT_NO_KEYRANGE"); return e && e[0] == '1'; }();

// rows given by `raw` (mode 1 or 2) -> sorted positions, distinct rows, run offsets
static int sort_dedup_run(const RawSrc& raw, int raw_mode, int64_t n, int64_t n_rows, int32_t* d_sorted_pos,
                          int32_t* d_sorted_slot, int64_t* d_uniq_rows, int32_t* d_seg_off, int32_t* d_n_uniq,
                          int32_t* d_inverse, void* ws, int32_t* err_flag, hipStream_t st) {
  RX_REQUIRE(((uintptr_t)ws & 15) == 0, "sort workspace must be 16-byte aligned");
  if (sort_keyrange_fits(raw_mode, n, n_rows) && !rx_sort_force_general && !rx_sort_no_keyrange) {
    const unsigned grid = (unsigned)rx_cdiv(n_rows, KR_KEYS);
    const size_t lds = sort_keyrange_lds(n, n_rows);
    static bool attr_set[64] = {};      // LDS beyond 64 KiB per block has to be asked for, per device
    int device = 0;
    RX_HIP(hipGetDevice(&device));
    if (device < 0 || device >= 64 || !attr_set[device]) {
      RX_HIP(hipFuncSetAttribute((const void*)sort_dedup_keyrange_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)sort_keyrange_lds(KR_MAX_N, KR_MAX_ROWS)));
      if (device >= 0 && device < 64) attr_set[device] = true;
    }
    hipLaunchKernelGGL(sort_dedup_keyrange_kernel<1>, dim3(grid), dim3(KR_THREADS), lds, st, raw, (int)n, n_rows,
                       d_sorted_pos, d_sorted_slot, d_uniq_rows, d_seg_off, d_n_uniq, d_inverse, err_flag);
    RX_LAUNCH_CHECK();
    return 0;
  }
  if (n <= SMALL_MAX && !rx_sort_force_general)
    return sort_dedup_small(raw, raw_mode, n, n_rows, d_sorted_pos, d_sorted_slot, d_uniq_rows, d_seg_off, d_n_uniq, d_inverse, err_flag, st);
  SortWs w = sort_ws_layout(ws, n);
  int passes, bits;
  sort_plan(n_rows, &passes, &bits);

  uint32_t *kin = w.ka, *kout = w.kb, *vin = w.va, *vout = w.vb;
  for (int p = 0; p < passes; ++p) {
    if (sort_pass(w, raw, p == 0 ? raw_mode : 0, n_rows, err_flag, kin, vin, kout, vout, n, p * bits, bits, true, st)) return 1;
    uint32_t* t = kin; kin = kout; kout = t;
    t = vin; vin = vout; vout = t;
  }
  const int ndb = (int)rx_cdiv(n, DEDUP_TILE);
  hipLaunchKernelGGL(dedup_count_kernel, dim3(ndb), dim3(RX_BLOCK), 0, st, kin, n, w.counts);
  RX_LAUNCH_CHECK();
  if (ndb <= SCAN_FOLD_MAX) {
    // few block counts: every block of the write kernel scans them itself
    hipLaunchKernelGGL(dedup_write_kernel<true>, dim3(ndb), dim3(RX_BLOCK), 0, st, kin, vin, n, w.counts, ndb, d_sorted_pos, d_uniq_rows, d_seg_off, d_n_uniq, d_inverse, d_sorted_slot);
    RX_LAUNCH_CHECK();
    return 0;
  }
  if (ndb <= (1 << 18)) {
    // few enough block counts for one block to scan in place
    hipLaunchKernelGGL(scan_totals_kernel, dim3(1), dim3(1024), 0, st, w.counts, (int64_t)ndb);
    RX_LAUNCH_CHECK();
  } else {
    // tiles in place + totals, then fold the totals back in
    const int n_tiles = (int)rx_cdiv(ndb, SCAN_TILE);
    hipLaunchKernelGGL(scan_tiles_kernel, dim3(n_tiles), dim3(RX_BLOCK), 0, st, w.counts, (int64_t)ndb, w.tiles);
    RX_LAUNCH_CHECK();
    hipLaunchKernelGGL(scan_totals_kernel, dim3(1), dim3(1024), 0, st, w.tiles, (int64_t)n_tiles);
    RX_LAUNCH_CHECK();
    hipLaunchKernelGGL(add_tile_offsets_kernel, dim3(n_tiles), dim3(RX_BLOCK), 0, st, w.counts, (int64_t)ndb, w.tiles);
    RX_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(dedup_write_kernel<false>, dim3(ndb), dim3(RX_BLOCK), 0, st, kin, vin, n, w.counts, ndb, d_sorted_pos, d_uniq_rows, d_seg_off, d_n_uniq, d_inverse, d_sorted_slot);
  RX_LAUNCH_CHECK();
  return 0;
}

extern "C" int rx_sort_dedup(const int64_t* d_idx, int64_t n, int64_t n_rows, int32_t* d_sorted_pos,
                             int32_t* d_sorted_slot, int64_t* d_uniq_rows, int32_t* d_seg_off,
                             int32_t* d_n_uniq, int32_t* d_inverse, void* ws, size_t ws_bytes,
                             int32_t* err_flag, int device, void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(n >= 0 && n < (1ll << 31), "rx_sort_dedup: n=%lld outside [0, 2^31)", (long long)n);
  RX_REQUIRE(n_rows >= 1 && n_rows <= (1ll << 32), "rx_sort_dedup: n_rows=%lld outside [1, 2^32]", (long long)n_rows);
  RX_REQUIRE(d_seg_off && d_n_uniq, "rx_sort_dedup: null output");
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) {
    RX_HIP(hipMemsetAsync(d_n_uniq, 0, sizeof(int32_t), st));
    RX_HIP(hipMemsetAsync(d_seg_off, 0, sizeof(int32_t), st));
    return 0;
  }
  RX_REQUIRE(d_idx && d_sorted_pos && d_uniq_rows && ws, "rx_sort_dedup: null argument");
  RX_REQUIRE(ws_bytes >= rx_sort_dedup_workspace_bytes(n), "rx_sort_dedup: workspace %zu < %zu bytes", ws_bytes, rx_sort_dedup_workspace_bytes(n));
  RawSrc raw = {};
  raw.idx = d_idx;
  return sort_dedup_run(raw, 1, n, n_rows, d_sorted_pos, d_sorted_slot, d_uniq_rows, d_seg_off, d_n_uniq, d_inverse, ws, err_flag, st);
}

extern "C" int rx_sort_dedup_fm(const rx_fm_args* a, int32_t* d_sorted_pos, int32_t* d_sorted_slot,
                                int64_t* d_uniq_rows, int32_t* d_seg_off, int32_t* d_n_uniq, int32_t* d_inverse,
                                void* ws, size_t ws_bytes, int32_t* err_flag, int device, void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(a && a->B >= 0 && a->F >= 1 && a->N >= 1, "rx_sort_dedup_fm: bad shape");
  RX_REQUIRE(a->ex_idx && a->ex_base && a->item_idx, "rx_sort_dedup_fm: null index pointer");
  RX_REQUIRE(a->ex_ld >= a->F && a->item_ld >= a->N, "rx_sort_dedup_fm: index row strides smaller than F / N");
  const int64_t n = a->B * ((int64_t)a->F + a->N);
  const int64_t n_rows = a->n_feat;
  RX_REQUIRE(n < (1ll << 31), "rx_sort_dedup_fm: n=%lld outside [0, 2^31)", (long long)n);
  RX_REQUIRE(n_rows >= 1 && n_rows <= (1ll << 32), "rx_sort_dedup_fm: n_rows=%lld outside [1, 2^32]", (long long)n_rows);
  RX_REQUIRE(d_seg_off && d_n_uniq, "rx_sort_dedup_fm: null output");
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) {
    RX_HIP(hipMemsetAsync(d_n_uniq, 0, sizeof(int32_t), st));
    RX_HIP(hipMemsetAsync(d_seg_off, 0, sizeof(int32_t), st));
    return 0;
  }
  RX_REQUIRE(d_sorted_pos && d_uniq_rows && ws, "rx_sort_dedup_fm: null argument");
  RX_REQUIRE(ws_bytes >= rx_sort_dedup_workspace_bytes(n), "rx_sort_dedup_fm: workspace %zu < %zu bytes", ws_bytes, rx_sort_dedup_workspace_bytes(n));
  RawSrc raw = {};
  raw.ex_idx = a->ex_idx;
  raw.ex_base = a->ex_base;
  raw.item_idx = a->item_idx;
  raw.ex_ld = a->ex_ld;
  raw.item_ld = a->item_ld;
  raw.item_base = a->item_base;
  raw.n_ex = (uint32_t)(a->B * a->F);
  raw.F = (uint32_t)a->F;
  raw.N = (uint32_t)a->N;
  return sort_dedup_run(raw, 2, n, n_rows, d_sorted_pos, d_sorted_slot, d_uniq_rows, d_seg_off, d_n_uniq, d_inverse, ws, err_flag, st);
}

// ---------------------------------------------------------------- bucketise by owning rank
// Row r lives on rank r % G.  A single stable radix pass keyed by the owner groups a rank's
// indices by destination while keeping their original order inside each group.
__global__ void __launch_bounds__(RX_BLOCK)
owner_init_kernel(const int64_t* __restrict__ idx, int64_t n, const int32_t* __restrict__ n_valid, int G,
                  int64_t n_rows_global, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals, int32_t* err_flag) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (n_valid && i >= n_valid[0]) {  // padding of a list whose length lives on the device: spare bin, sent nowhere
    keys[i] = (uint32_t)G;
    vals[i] = (uint32_t)i;
    return;
  }
  int64_t r = idx[i];
  if (r < 0 || r >= n_rows_global) {
    if (err_flag) atomicOr(err_flag, RX_ERR_INDEX);
    r = r < 0 ? 0 : n_rows_global - 1;
  }
  keys[i] = (uint32_t)(r % G);
  vals[i] = (uint32_t)i;
}

__global__ void __launch_bounds__(RX_BLOCK)
owner_finish_kernel(const int64_t* __restrict__ idx, const uint32_t* __restrict__ perm_in, int64_t n, int G,
                    int64_t n_rows_global, const uint32_t* __restrict__ scanned_hist,
                    const uint32_t* __restrict__ tile_off, int n_blocks, int64_t* __restrict__ counts,
                    int64_t* __restrict__ send_local, int32_t* __restrict__ perm) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < G) {
    // start of digit d = scanned count entry (d, block 0); bin G exists and is empty
    const int64_t e0 = j * n_blocks, e1 = (j + 1) * n_blocks;
    const int64_t lo = (int64_t)scanned_hist[e0] + tile_off[e0 / SCAN_TILE];
    const int64_t hi = (int64_t)scanned_hist[e1] + tile_off[e1 / SCAN_TILE];
    counts[j] = hi - lo;
  }
  if (j >= n) return;
  const uint32_t p = perm_in[j];
  int64_t r = idx[p];
  if (r < 0) r = 0;
  if (r >= n_rows_global) r = n_rows_global - 1;
  send_local[j] = r / G;
  perm[j] = (int32_t)p;
}

extern "C" size_t rx_bucketize_workspace_bytes(int64_t n, int32_t G) {
  (void)G;
  return rx_sort_dedup_workspace_bytes(n);
}

extern "C" int rx_bucketize_by_owner(const int64_t* d_idx, int64_t n, const int32_t* d_n_valid, int32_t G,
                                     int64_t n_rows_global, int64_t* d_counts, int64_t* d_send_local,
                                     int32_t* d_perm, void* ws, size_t ws_bytes, int32_t* err_flag, int device,
                                     void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(G >= 1 && G < SORT_MAX_BINS, "rx_bucketize_by_owner: G=%d outside 1..%d", G, SORT_MAX_BINS - 1);
  RX_REQUIRE(n >= 0 && n < (1ll << 31) && n_rows_global >= 1, "rx_bucketize_by_owner: bad sizes");
  RX_REQUIRE(d_counts, "rx_bucketize_by_owner: null counts");
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) {
    RX_HIP(hipMemsetAsync(d_counts, 0, sizeof(int64_t) * G, st));
    return 0;
  }
  RX_REQUIRE(d_idx && d_send_local && d_perm && ws, "rx_bucketize_by_owner: null argument");
  RX_REQUIRE(ws_bytes >= rx_bucketize_workspace_bytes(n, G), "rx_bucketize_by_owner: workspace too small");
  SortWs w = sort_ws_layout(ws, n);
  const int nb = (int)rx_cdiv(n, SORT_TILE);
  int bits = 1;
  while ((1 << bits) < G + 1) ++bits;  // one spare bin so that bin G marks the end of bin G-1
  hipLaunchKernelGGL(owner_init_kernel, dim3((unsigned)rx_cdiv(n, RX_BLOCK)), dim3(RX_BLOCK), 0, st, d_idx, n, d_n_valid, G, n_rows_global, w.ka, w.va, err_flag);
  RX_LAUNCH_CHECK();
  if (sort_pass(w, RawSrc{}, 0, n_rows_global, err_flag, w.ka, w.va, w.kb, w.vb, n, 0, bits, false, st)) return 1;
  const int64_t work = n > G ? n : G;
  hipLaunchKernelGGL(owner_finish_kernel, dim3((unsigned)rx_cdiv(work, RX_BLOCK)), dim3(RX_BLOCK), 0, st, d_idx, w.vb, n, G, n_rows_global, w.hist, w.tiles, nb, d_counts, d_send_local, d_perm);
  RX_LAUNCH_CHECK();
  return 0;
}
