// Field-grouped backward of the one-item-per-sample field models (gfx950).
//
// A batch of the multi-field FM family names, per sample, one row in each of n_fields disjoint row
// ranges of one table.  The occurrences of a field therefore share nothing with those of another
// field, and a field holds exactly B of them.  rx_fg_group orders each field's occurrences by
// (bin, row >> 8, sample) with bin = row for the field's first 256 rows and 256 + (row & 255) for
// the others: rows are grouped, the samples of a row ascend, and a row's run never leaves its field.
// The first 256 rows of a field (where frequency-sorted vocabularies keep their hot ids) get a bin
// each, so a hot row never shares a bin with rows that still have to be told apart; the order is
// valid whatever the distribution.  It needs one stable counting pass over the whole field and then
// only a small sort inside each of the 256 shared bins, instead of a full-width radix sort over all
// occurrences of all fields:
//   K1 hist     a block reads a [256 samples, n_fields] slab of the index matrix with one coalesced
//               sweep, checks the indices, counts the bins per field in LDS and leaves the field-local
//               rows transposed ([field][sample], uint32)
//   K2 scatter  block (tile of 4096 samples, field): bin starts from the field's tile counts, a
//               stable rank per key from 64-lane matches and per-wave counters, coalesced write-out.
//               Keys of a single-row bin go straight to the final arrays; the others are written as
//               (row >> 8, sample) packed in one word.  Packed words of one bin are distinct and
//               ascend with (row >> 8, sample), so sorting them as plain integers is the stable sort.
//   K3 bins     a wavefront per shared bin: up to 512 words in registers, LSD passes of 4 bits ranked
//               with 64-lane matches, no block barrier.  Larger bins (a hot row beyond the first 256)
//               go to a 1024-thread launch with up to 16384 words per block; anything larger than
//               that is sorted through global memory.
// The packed word needs bits(B - 1) + bits((largest field - 1) >> 8) bits: 32-bit words serve tables
// of up to 16.7 M rows per field at B = 65536; wider fields (the 125 M-row tables) take the same
// kernels instantiated on 64-bit words (WIDE), with half as many words per LDS-resident big bin.
// No float or order-dependent atomic is involved: counts are integers, ranks come from prefix sums.
//
// rx_fg_reduce then sums, for every run, g_b T_b (+ the tower-input gradient slices) with one lane
// group per POSITION: a segmented scan by row over the lane groups of a wavefront, a carry from
// step to step, partial sums between wavefront windows joined by a second launch.  The tree shape
// depends only on the positions, so results are bitwise reproducible.
#include "rx_common.h"

#define FG_TS 4096          // samples per scatter tile
#define FG_ITEMS 16         // keys per thread in the scatter
#define FG_HB 256           // samples per histogram block
#define FG_BINS 512         // 256 single-row bins + 256 shared bins per field
#define FG_WAVE_CAP 512     // words a wavefront sorts in registers
#define FG_WAVE_ROUNDS (FG_WAVE_CAP / 64)
#define FG_BIG_CAP 16384    // words the 1024-thread kernel sorts in LDS
#define FG_BIG_ROUNDS (FG_BIG_CAP / 1024)
#define FG_BIG_CAP_WIDE 7168 // the same with 64-bit words (56 KB)
#define FG_BIG_MAX 128      // bins above FG_WAVE_CAP per field (B <= 65536)
#define FG_BIG_GRID 8       // blocks per field of the big-bin launch
#define FG_MAX_B 65536

struct FgPlan {
  const int64_t* ex_idx;
  const int64_t* item_idx;
  int64_t ex_ld, item_ld;
  int B, nf, NT, BB;                      // samples, fields, scatter tiles per field, bits of a sample id
  uint32_t base[RX_FG_MAX_FIELDS + 1];    // first table row of each field; base[nf] = rows of the table
};

__device__ __forceinline__ int fg_high_bits(uint32_t card) {  // bits of (row >> 8) for rows < card
  return card > 256u ? 32 - __clz((int)((card - 1u) >> 8)) : 0;
}

__device__ __forceinline__ uint32_t fg_bin(uint32_t row) { return row < 256u ? row : 256u + (row & 255u); }

// exclusive scan over the block (256 or 1024 threads); returns this thread's prefix, *total = sum
template <typename T, int NTHREADS>
__device__ __forceinline__ T fg_block_scan(T v, T* lds_w, T* total) {
  constexpr int NW = NTHREADS / 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  T inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const T t = __shfl_up(inc, off, 64);
    if (lane >= off) inc += t;
  }
  if (lane == 63) lds_w[wave] = inc;
  __syncthreads();
  T base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    const T x = lds_w[w];
    if (w < wave) base += x;
    tot += x;
  }
  if (total) *total = tot;
  __syncthreads();
  return base + inc - v;
}

// 64-lane match on a BITS-wide digit: per bit one ballot; sb is all ones where the lane's bit is
// set, so (ballot ^ sb) marks the lanes whose bit differs from mine.
template <int BITS>
__device__ __forceinline__ unsigned long long fg_match(uint32_t d, bool live) {
  unsigned long long p = __ballot(live);
#pragma unroll
  for (int bit = 0; bit < BITS; ++bit) {
    const int32_t sb = (int32_t)(d << (31 - bit)) >> 31;
    const unsigned long long m = __ballot(sb != 0);
    p &= ~(m ^ (unsigned long long)(long long)sb);
  }
  return live ? p : 0ull;
}

template <bool WIDE> struct FgWord { using type = uint32_t; };
template <> struct FgWord<true> { using type = uint64_t; };

// ---------------------------------------------------------------- K1: histogram + transpose
// Bin counts of a block fit 16 bits (256 samples), two per LDS word.
#define FG_HIST_ROUND 16
__global__ void __launch_bounds__(RX_BLOCK)
fg_hist_kernel(FgPlan pl, uint32_t* __restrict__ keyT, uint32_t* __restrict__ hcnt, int32_t* err_flag) {
  extern __shared__ uint32_t fg_lds[];  // hist [nf][FG_BINS / 2] | stage [nf][257]
  const int nf = pl.nf;
  uint32_t* hist = fg_lds;
  uint32_t* stage = fg_lds + nf * (FG_BINS / 2);
  for (int i = threadIdx.x; i < nf * (FG_BINS / 2); i += RX_BLOCK) hist[i] = 0;
  __syncthreads();
  const int b0 = blockIdx.x * FG_HB;
  const int rows_here = pl.B - b0 < FG_HB ? pl.B - b0 : FG_HB;
  const int n_here = rows_here * nf;
  bool bad = false;
  for (int e0 = 0; e0 < n_here; e0 += FG_HIST_ROUND * RX_BLOCK) {
    int64_t r[FG_HIST_ROUND];
#pragma unroll
    for (int j = 0; j < FG_HIST_ROUND; ++j) {  // all loads of a round are issued before any is used
      int e = e0 + j * RX_BLOCK + threadIdx.x;
      e = e < n_here ? e : n_here - 1;
      const int s = e / nf, f = e - s * nf;
      const int64_t* p = f < nf - 1 ? pl.ex_idx + (int64_t)(b0 + s) * pl.ex_ld + f : pl.item_idx + (int64_t)(b0 + s) * pl.item_ld;
      r[j] = *p;
    }
#pragma unroll
    for (int j = 0; j < FG_HIST_ROUND; ++j) {
      const int e = e0 + j * RX_BLOCK + threadIdx.x;
      if (e < n_here) {
        const int s = e / nf, f = e - s * nf;
        const int64_t card = (int64_t)pl.base[f + 1] - (int64_t)pl.base[f];
        int64_t v = r[j];
        if (v < 0 || v >= card) {
          bad = true;
          v = v < 0 ? 0 : card - 1;
        }
        stage[f * 257 + s] = (uint32_t)v;
        const uint32_t bin = fg_bin((uint32_t)v);
        atomicAdd(&hist[f * (FG_BINS / 2) + (bin >> 1)], 1u << (16 * (bin & 1u)));
      }
    }
  }
  if (bad && err_flag) atomicOr(err_flag, RX_ERR_INDEX);
  __syncthreads();
  // this block's packed counts, one coalesced row per field; fg_tile_counts adds the blocks of a tile
  for (int i = threadIdx.x; i < nf * (FG_BINS / 2); i += RX_BLOCK) {
    const int f = i / (FG_BINS / 2), w = i - f * (FG_BINS / 2);
    hcnt[((size_t)f * gridDim.x + blockIdx.x) * (FG_BINS / 2) + w] = hist[i];
  }
  if ((int)threadIdx.x < rows_here)
    for (int f = 0; f < nf; ++f) keyT[(size_t)f * pl.B + b0 + threadIdx.x] = stage[f * 257 + threadIdx.x];
}

// K1b: counts[field][tile][bin] = sum of the packed counts of the tile's histogram blocks
__global__ void __launch_bounds__(RX_BLOCK)
fg_tile_counts_kernel(FgPlan pl, const uint32_t* __restrict__ hcnt, int n_hist_blocks, uint32_t* __restrict__ counts) {
  const int t = blockIdx.x, f = blockIdx.y, w = threadIdx.x;
  constexpr int PER = FG_TS / FG_HB;
  uint32_t lo = 0, hi = 0;
  uint32_t x[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int hb = t * PER + k;
    x[k] = hb < n_hist_blocks ? hcnt[((size_t)f * n_hist_blocks + hb) * (FG_BINS / 2) + w] : 0u;
  }
#pragma unroll
  for (int k = 0; k < PER; ++k) { lo += x[k] & 0xFFFFu; hi += x[k] >> 16; }
  reinterpret_cast<uint2*>(counts + ((size_t)f * pl.NT + t) * FG_BINS)[w] = make_uint2(lo, hi);
}

// ---------------------------------------------------------------- K2: stable scatter by bin
template <bool WIDE>
__global__ void __launch_bounds__(RX_BLOCK)
fg_scatter_kernel(FgPlan pl, const uint32_t* __restrict__ keyT, const uint32_t* __restrict__ counts,
                  void* __restrict__ part_v, uint32_t* __restrict__ bstart, uint32_t* __restrict__ biglist,
                  uint32_t* __restrict__ srow, uint32_t* __restrict__ sb) {
  using W = typename FgWord<WIDE>::type;
  W* __restrict__ part = reinterpret_cast<W*>(part_v);

  __shared__ uint32_t gstart[FG_BINS], texcl[FG_BINS];
  __shared__ uint32_t stg[2 * FG_TS];  // first the 4 x FG_BINS wave counters, later the staged (key, sample) pairs
  __shared__ uint32_t lds4[4];
  __shared__ uint32_t big_n;
  const int t = blockIdx.x, f = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int B = pl.B;
  uint32_t* wave_cnt = stg;
  for (int i = threadIdx.x; i < 4 * FG_BINS; i += RX_BLOCK) wave_cnt[i] = 0;
  if (threadIdx.x == 0) big_n = 0;
  // keys of this tile: wave w owns 1024 consecutive samples, lane-strided
  const int tb = t * FG_TS;
  const int wb = tb + wave * (64 * FG_ITEMS);
  uint32_t key[FG_ITEMS];
#pragma unroll
  for (int it = 0; it < FG_ITEMS; ++it) {
    const int i = wb + it * 64 + lane;
    key[it] = keyT[(size_t)f * B + (i < B ? i : B - 1)];
  }
  // bin starts: thread d sums the counts of bins 2d, 2d + 1 over the field's tiles
  {
    const int d = threadIdx.x;
    uint32_t tot0 = 0, tot1 = 0, bef0 = 0, bef1 = 0;
    const uint2* c = reinterpret_cast<const uint2*>(counts + (size_t)f * pl.NT * FG_BINS) + d;
    for (int tt = 0; tt < pl.NT; ++tt) {
      const uint2 x = c[(size_t)tt * (FG_BINS / 2)];
      tot0 += x.x; tot1 += x.y;
      bef0 += tt < t ? x.x : 0u; bef1 += tt < t ? x.y : 0u;
    }
    const uint32_t dstart = fg_block_scan<uint32_t, RX_BLOCK>(tot0 + tot1, lds4, nullptr);
    gstart[2 * d] = dstart + bef0;
    gstart[2 * d + 1] = dstart + tot0 + bef1;
    if (t == 0) {
      bstart[f * (FG_BINS + 1) + 2 * d] = dstart;
      bstart[f * (FG_BINS + 1) + 2 * d + 1] = dstart + tot0;
      if (d == RX_BLOCK - 1) bstart[f * (FG_BINS + 1) + FG_BINS] = (uint32_t)B;
      if (d >= 128) {  // shared bins too large for one wavefront
        if (tot0 > FG_WAVE_CAP) { const uint32_t slot = atomicAdd(&big_n, 1u); if (slot < FG_BIG_MAX) biglist[f * (FG_BIG_MAX + 1) + 1 + slot] = 2u * d; }
        if (tot1 > FG_WAVE_CAP) { const uint32_t slot = atomicAdd(&big_n, 1u); if (slot < FG_BIG_MAX) biglist[f * (FG_BIG_MAX + 1) + 1 + slot] = 2u * d + 1u; }
      }
    }
  }
  __syncthreads();
  if (t == 0 && threadIdx.x == 0) biglist[f * (FG_BIG_MAX + 1)] = big_n < FG_BIG_MAX ? big_n : FG_BIG_MAX;
  // rank inside the wave's 1024 keys: all matches first (independent ballot chains), then the
  // counters key by key (LDS operations of a wavefront execute in program order)
  uint32_t* my_cnt = wave_cnt + wave * FG_BINS;
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  unsigned long long peers[FG_ITEMS];
  uint32_t rank[FG_ITEMS];
#pragma unroll
  for (int it = 0; it < FG_ITEMS; ++it) peers[it] = fg_match<9>(fg_bin(key[it]), wb + it * 64 + lane < B);
#pragma unroll
  for (int it = 0; it < FG_ITEMS; ++it) {
    const uint32_t d = fg_bin(key[it]);
    const unsigned long long p = peers[it];
    uint32_t base = 0;
    if (p) base = my_cnt[d];
    if (p && lane == __ffsll((long long)p) - 1) my_cnt[d] = base + (uint32_t)__popcll(p);
    rank[it] = base + (uint32_t)__popcll(p & lt_mask);
  }
  __syncthreads();
  {
    const int d = threadIdx.x;
    uint32_t run[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      run[h] = 0;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const uint32_t c = wave_cnt[w * FG_BINS + 2 * d + h];
        wave_cnt[w * FG_BINS + 2 * d + h] = run[h];
        run[h] += c;
      }
    }
    const uint32_t ex = fg_block_scan<uint32_t, RX_BLOCK>(run[0] + run[1], lds4, nullptr);
    texcl[2 * d] = ex;
    texcl[2 * d + 1] = ex + run[0];
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < FG_ITEMS; ++it) {
    const uint32_t d = fg_bin(key[it]);
    rank[it] += texcl[d] + my_cnt[d];
  }
  __syncthreads();
  uint32_t* sk = stg;
  uint32_t* sv = stg + FG_TS;
#pragma unroll
  for (int it = 0; it < FG_ITEMS; ++it) {
    const int i = wb + it * 64 + lane;
    if (i < B) {
      sk[rank[it]] = key[it];
      sv[rank[it]] = (uint32_t)i;
    }
  }
  __syncthreads();
  const int tile_n = tb + FG_TS <= B ? FG_TS : B - tb;
  const uint32_t fbase = pl.base[f];
#pragma unroll
  for (int it = 0; it < FG_ITEMS; ++it) {
    const int j = it * RX_BLOCK + threadIdx.x;
    if (j < tile_n) {
      const uint32_t k = sk[j];
      const uint32_t d = fg_bin(k);
      const size_t dst = (size_t)f * B + gstart[d] + ((uint32_t)j - texcl[d]);
      if (d < 256u) {  // single-row bin: final arrays, nothing left to sort
        srow[dst] = fbase + k;
        sb[dst] = sv[j];
      } else {
        part[dst] = ((W)(k >> 8) << pl.BB) | (W)sv[j];
      }
    }
  }
}

// ---------------------------------------------------------------- K3: sort inside the shared bins
// One wavefront per bin, its words in registers (round r: word r * 64 + lane).  Per 8-bit pass:
// every round matches its digits, the group leaders count; the lanes turn the 256 counts (four each)
// into starts; the rounds then hand out positions in order (leaders advance the starts), the words
// go to the wavefront's LDS strip at their new positions and are read back in order.
template <bool WIDE>
__global__ void __launch_bounds__(RX_BLOCK)
fg_bin_wave_kernel(FgPlan pl, const void* __restrict__ part_v, const uint32_t* __restrict__ bstart,
                   uint32_t* __restrict__ srow, uint32_t* __restrict__ sb) {
  using W = typename FgWord<WIDE>::type;
  const W* __restrict__ part = reinterpret_cast<const W*>(part_v);
  __shared__ W strip[RX_BLOCK / 64][FG_WAVE_CAP];
  __shared__ __align__(16) uint32_t cnt[RX_BLOCK / 64][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wid = blockIdx.x * (RX_BLOCK / 64) + wave;   // (field, shared bin)
  const int f = wid >> 8, d = wid & 255;
  if (f >= pl.nf) return;
  const int hb = fg_high_bits(pl.base[f + 1] - pl.base[f]);
  if (hb == 0) return;
  const uint32_t lo = bstart[f * (FG_BINS + 1) + 256 + d], hi = bstart[f * (FG_BINS + 1) + 256 + d + 1];
  const int m = (int)(hi - lo);
  if (m == 0 || m > FG_WAVE_CAP) return;
  const size_t p0 = (size_t)f * pl.B + lo;
  W x[FG_WAVE_ROUNDS];
#pragma unroll
  for (int r = 0; r < FG_WAVE_ROUNDS; ++r) {
    const int j = r * 64 + lane;
    x[r] = part[p0 + (j < m ? j : m - 1)];
  }
  W* buf = strip[wave];
  uint32_t* c = cnt[wave];
  uint4* c4 = reinterpret_cast<uint4*>(c);
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  for (int bit = pl.BB; bit < pl.BB + hb; bit += 8) {
    const uint32_t mask = (1u << (pl.BB + hb - bit < 8 ? pl.BB + hb - bit : 8)) - 1u;
    c4[lane] = make_uint4(0u, 0u, 0u, 0u);
    unsigned long long peers[FG_WAVE_ROUNDS];
#pragma unroll
    for (int r = 0; r < FG_WAVE_ROUNDS; ++r) {
      if (r * 64 < m) {
        const uint32_t dg = (uint32_t)(x[r] >> bit) & mask;
        peers[r] = fg_match<8>(dg, r * 64 + lane < m);
        if (peers[r] && lane == __ffsll((long long)peers[r]) - 1) c[dg] += (uint32_t)__popcll(peers[r]);
      }
    }
    {  // exclusive scan of the 256 counts: lane l holds counts 4l .. 4l + 3
      const uint4 q = c4[lane];
      const uint32_t v = q.x + q.y + q.z + q.w;
      uint32_t inc = v;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(inc, off, 64);
        if (lane >= off) inc += t;
      }
      const uint32_t ex = inc - v;
      c4[lane] = make_uint4(ex, ex + q.x, ex + q.x + q.y, ex + q.x + q.y + q.z);
    }
#pragma unroll
    for (int r = 0; r < FG_WAVE_ROUNDS; ++r) {
      if (r * 64 < m) {
        const uint32_t dg = (uint32_t)(x[r] >> bit) & mask;
        const unsigned long long p = peers[r];
        uint32_t base = 0;
        if (p) base = c[dg];
        if (p && lane == __ffsll((long long)p) - 1) c[dg] = base + (uint32_t)__popcll(p);
        if (p) buf[base + (uint32_t)__popcll(p & lt_mask)] = x[r];
      }
    }
#pragma unroll
    for (int r = 0; r < FG_WAVE_ROUNDS; ++r)
      if (r * 64 < m) x[r] = buf[r * 64 + lane < m ? r * 64 + lane : m - 1];
  }
  const uint32_t bmask = (1u << pl.BB) - 1u;
  const uint32_t rbase = pl.base[f] + (uint32_t)d;
#pragma unroll
  for (int r = 0; r < FG_WAVE_ROUNDS; ++r) {
    const int j = r * 64 + lane;
    if (j < m) {
      srow[p0 + j] = rbase + ((uint32_t)(x[r] >> pl.BB) << 8);
      sb[p0 + j] = (uint32_t)x[r] & bmask;
    }
  }
}

// Stable LSD radix sort of m words through memory (`a`, scratch `b`), bits [lo, lo + nbits), two
// bits per pass: every thread owns k consecutive words, counts its digits, a block scan over the
// packed counts gives each thread the start of its words per digit.  The buffers are global memory
// shared by the block's waves, so accesses bypass the per-CU cache and stores are drained before
// each barrier.  Only bins beyond FG_BIG_CAP words come here.
template <typename W> __device__ __forceinline__ W fg_gld(const W* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename W> __device__ __forceinline__ void fg_gst(W* p, W v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int NTHREADS, typename W>
__device__ __forceinline__ W* fg_radix_sort_global(W* a, W* b, int m, int lo, int nbits,
                                                   unsigned long long* lds_w) {
  const int k = (m + NTHREADS - 1) / NTHREADS;
  const int j0 = threadIdx.x * k < m ? threadIdx.x * k : m;
  const int j1 = j0 + k < m ? j0 + k : m;
  for (int bit = lo; bit < lo + nbits; bit += 2) {
    const uint32_t mask = (lo + nbits - bit) >= 2 ? 3u : 1u;
    unsigned long long c = 0;  // counts of digits 0..2 in 21-bit fields (digit 3 follows from the thread's length)
    for (int j = j0; j < j1; ++j) {
      const uint32_t d = (uint32_t)(fg_gld(a + j) >> bit) & mask;
      c += d < 3 ? 1ull << (21 * d) : 0ull;
    }
    unsigned long long tot;
    const unsigned long long pre = fg_block_scan<unsigned long long, NTHREADS>(c, lds_w, &tot);
    const uint32_t t0 = (uint32_t)(tot & 0x1FFFFF), t1 = (uint32_t)((tot >> 21) & 0x1FFFFF), t2 = (uint32_t)((tot >> 42) & 0x1FFFFF);
    const uint32_t p0 = (uint32_t)(pre & 0x1FFFFF), p1 = (uint32_t)((pre >> 21) & 0x1FFFFF), p2 = (uint32_t)((pre >> 42) & 0x1FFFFF);
    uint32_t at0 = p0, at1 = t0 + p1, at2 = t0 + t1 + p2, at3 = t0 + t1 + t2 + ((uint32_t)j0 - p0 - p1 - p2);
    for (int j = j0; j < j1; ++j) {
      const W x = fg_gld(a + j);
      const uint32_t d = (uint32_t)(x >> bit) & mask;
      const uint32_t pos = d == 0 ? at0 : (d == 1 ? at1 : (d == 2 ? at2 : at3));
      at0 += d == 0; at1 += d == 1; at2 += d == 2; at3 += d == 3;
      fg_gst(b + pos, x);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    W* t = a; a = b; b = t;
  }
  return a;
}

// Bins above FG_WAVE_CAP words: the same 4-bit passes with a whole 1024-thread block per bin (wave w
// owns a contiguous chunk, so (wave, round, lane) is the original order); blocks walk their field's
// list of such bins.
template <bool WIDE>
__global__ void __launch_bounds__(1024)
fg_bin_big_kernel(FgPlan pl, void* __restrict__ part_v, void* __restrict__ spare_v,
                  const uint32_t* __restrict__ bstart, const uint32_t* __restrict__ biglist,
                  uint32_t* __restrict__ srow, uint32_t* __restrict__ sb) {
  using W = typename FgWord<WIDE>::type;
  constexpr int CAP = WIDE ? FG_BIG_CAP_WIDE : FG_BIG_CAP;     // words a block sorts in LDS (56 KB / 64 KB)
  constexpr int ROUNDS = CAP / 1024;
  W* __restrict__ part = reinterpret_cast<W*>(part_v);
  W* __restrict__ spare = reinterpret_cast<W*>(spare_v);
  __shared__ W buf[CAP];
  __shared__ uint32_t cnt[16][16];
  __shared__ uint32_t lds16[16];
  __shared__ unsigned long long lds_w[16];
  const int f = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n_big = (int)biglist[f * (FG_BIG_MAX + 1)];
  const int hb = fg_high_bits(pl.base[f + 1] - pl.base[f]);
  const uint32_t bmask = (1u << pl.BB) - 1u;
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  for (int e = blockIdx.x; e < n_big; e += FG_BIG_GRID) {
    const int bin = (int)biglist[f * (FG_BIG_MAX + 1) + 1 + e];
    const uint32_t lo = bstart[f * (FG_BINS + 1) + bin], hi = bstart[f * (FG_BINS + 1) + bin + 1];
    const int m = (int)(hi - lo);
    const size_t p0 = (size_t)f * pl.B + lo;
    const uint32_t rbase = pl.base[f] + (uint32_t)(bin - 256);
    if (m > CAP) {
      const W* res = fg_radix_sort_global<1024, W>(part + p0, spare + p0, m, pl.BB, hb, lds_w);
      for (int j = threadIdx.x; j < m; j += 1024) {
        const W x = fg_gld(res + j);
        srow[p0 + j] = rbase + ((uint32_t)(x >> pl.BB) << 8);
        sb[p0 + j] = (uint32_t)x & bmask;
      }
      __syncthreads();
      continue;
    }
    const int ch = ((m + 15) / 16 + 63) / 64 * 64;   // words per wave, whole rounds
    const int w0 = wave * ch;
    W x[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const int j = w0 + r * 64 + lane;
      x[r] = part[p0 + (j < m ? j : m - 1)];
    }
    for (int bit = pl.BB; bit < pl.BB + hb; bit += 4) {
      const uint32_t mask = (1u << (pl.BB + hb - bit < 4 ? pl.BB + hb - bit : 4)) - 1u;
      if (lane < 16) cnt[wave][lane] = 0;
      unsigned long long peers[ROUNDS];
#pragma unroll
      for (int r = 0; r < ROUNDS; ++r) {
        peers[r] = 0ull;
        if (r * 64 < ch) {
          const uint32_t dg = (uint32_t)(x[r] >> bit) & mask;
          peers[r] = fg_match<4>(dg, w0 + r * 64 + lane < m);
          if (peers[r] && lane == __ffsll((long long)peers[r]) - 1) cnt[wave][dg] += (uint32_t)__popcll(peers[r]);
        }
      }
      __syncthreads();
      {  // starts in (digit, wave) order: thread t < 256 holds the count of digit t / 16 in wave t % 16
        const int tt = threadIdx.x;
        const uint32_t v = tt < 256 ? cnt[tt & 15][tt >> 4] : 0u;
        const uint32_t ex = fg_block_scan<uint32_t, 1024>(v, lds16, nullptr);
        if (tt < 256) cnt[tt & 15][tt >> 4] = ex;
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < ROUNDS; ++r) {
        if (r * 64 < ch) {
          const uint32_t dg = (uint32_t)(x[r] >> bit) & mask;
          const unsigned long long p = peers[r];
          uint32_t base = 0;
          if (p) base = cnt[wave][dg];
          if (p && lane == __ffsll((long long)p) - 1) cnt[wave][dg] = base + (uint32_t)__popcll(p);
          if (p) buf[base + (uint32_t)__popcll(p & lt_mask)] = x[r];
        }
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < ROUNDS; ++r) {
        const int j = w0 + r * 64 + lane;
        if (r * 64 < ch) x[r] = buf[j < m ? j : m - 1];
      }
      __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const int j = w0 + r * 64 + lane;
      if (r * 64 < ch && j < m) {
        srow[p0 + j] = rbase + ((uint32_t)(x[r] >> pl.BB) << 8);
        sb[p0 + j] = (uint32_t)x[r] & bmask;
      }
    }
  }
}

struct FgWs {
  uint32_t *keyT, *counts, *hcnt, *bstart, *biglist;
  void *part, *spare;     // packed words (4 or 8 bytes each) and the scratch of the through-memory sort
  size_t bytes;
};

static FgWs fg_ws_layout(void* ws, int64_t B, int nf) {
  FgWs w;
  const int64_t bb = B > 0 ? B : 1;
  const int NT = (int)rx_cdiv(bb, FG_TS);
  char* p = (char*)ws;
  w.counts = (uint32_t*)p; p += rx_align_up((size_t)nf * NT * FG_BINS * 4, 256);
  w.hcnt = (uint32_t*)p; p += rx_align_up((size_t)nf * rx_cdiv(bb, FG_HB) * (FG_BINS / 2) * 4, 256);
  w.keyT = (uint32_t*)p; p += rx_align_up((size_t)nf * bb * 4, 256);
  w.part = (void*)p; p += rx_align_up((size_t)nf * bb * 8, 256);
  w.spare = (void*)p; p += rx_align_up((size_t)nf * bb * 8, 256);
  w.bstart = (uint32_t*)p; p += rx_align_up((size_t)nf * (FG_BINS + 1) * 4, 256);
  w.biglist = (uint32_t*)p; p += rx_align_up((size_t)nf * (FG_BIG_MAX + 1) * 4, 256);
  w.bytes = (size_t)(p - (char*)ws);
  return w;
}

extern "C" size_t rx_fg_workspace_bytes(int64_t B, int32_t n_fields) {
  if (n_fields < 1 || n_fields > RX_FG_MAX_FIELDS) return 0;
  return fg_ws_layout(nullptr, B, n_fields).bytes;
}

static int fg_bits(uint64_t x) {  // bits needed for values 0..x
  int b = 0;
  while (x) { ++b; x >>= 1; }
  return b;
}

extern "C" int rx_fg_supported(const rx_fg_args* a) {
  if (!a || a->n_fields < 1 || a->n_fields > RX_FG_MAX_FIELDS || a->B < 1 || a->B > FG_MAX_B) return 0;
  int64_t max_card = 0;
  for (int f = 0; f < a->n_fields; ++f) {
    const int64_t card = a->field_base[f + 1] - a->field_base[f];
    if (a->field_base[f] < 0 || card < 1) return 0;
    if (card > max_card) max_card = card;
  }
  if (a->field_base[a->n_fields] > (1ll << 32) - 16) return 0;
  const int BB = fg_bits((uint64_t)(a->B - 1)) > 0 ? fg_bits((uint64_t)(a->B - 1)) : 1;
  const int hb = max_card > 256 ? fg_bits((uint64_t)((max_card - 1) >> 8)) : 0;
  return BB + hb <= 48 ? 1 : 0;
}

extern "C" int rx_fg_group(const rx_fg_args* a, uint32_t* d_srow, uint32_t* d_sb, void* ws, size_t ws_bytes,
                           int32_t* err_flag, int device, void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(a, "rx_fg_group: null args");
  RX_REQUIRE(a->n_fields >= 1 && a->n_fields <= RX_FG_MAX_FIELDS, "rx_fg_group: n_fields=%d outside 1..%d", a->n_fields, RX_FG_MAX_FIELDS);
  RX_REQUIRE(a->B >= 0 && a->B <= FG_MAX_B, "rx_fg_group: B=%lld outside 0..%d", (long long)a->B, FG_MAX_B);
  if (a->B == 0) return 0;
  RX_REQUIRE(rx_fg_supported(a), "rx_fg_group: field ranges must be ascending, non-empty and below 2^32 - 16 rows");
  RX_REQUIRE(a->ex_idx && a->item_idx && d_srow && d_sb && ws, "rx_fg_group: null pointer");
  RX_REQUIRE(a->ex_ld >= a->n_fields - 1 && a->item_ld >= 1, "rx_fg_group: index row strides too small");
  RX_REQUIRE(ws_bytes >= rx_fg_workspace_bytes(a->B, a->n_fields) && ((uintptr_t)ws & 255) == 0, "rx_fg_group: workspace too small or not 256-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  FgWs w = fg_ws_layout(ws, a->B, a->n_fields);
  FgPlan pl{};
  pl.ex_idx = a->ex_idx; pl.item_idx = a->item_idx; pl.ex_ld = a->ex_ld; pl.item_ld = a->item_ld;
  pl.B = (int)a->B; pl.nf = a->n_fields; pl.NT = (int)rx_cdiv(a->B, FG_TS);
  pl.BB = fg_bits((uint64_t)(a->B - 1)) > 0 ? fg_bits((uint64_t)(a->B - 1)) : 1;
  for (int f = 0; f <= a->n_fields; ++f) pl.base[f] = (uint32_t)a->field_base[f];
  const size_t lds1 = (size_t)pl.nf * (FG_BINS / 2 + 257) * sizeof(uint32_t);
  static bool attr_set[64] = {};
  if (lds1 > 64 * 1024 && (device >= 64 || !attr_set[device])) {  // LDS beyond 64 KiB per block has to be asked for, per device
    RX_HIP(hipFuncSetAttribute((const void*)fg_hist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, RX_FG_MAX_FIELDS * (FG_BINS / 2 + 257) * 4));
    if (device < 64) attr_set[device] = true;
  }
  const int n_hist = (int)rx_cdiv(a->B, FG_HB);
  hipLaunchKernelGGL(fg_hist_kernel, dim3(n_hist), dim3(RX_BLOCK), lds1, st, pl, w.keyT, w.hcnt, err_flag);
  RX_LAUNCH_CHECK();
  hipLaunchKernelGGL(fg_tile_counts_kernel, dim3(pl.NT, pl.nf), dim3(RX_BLOCK), 0, st, pl, w.hcnt, n_hist, w.counts);
  RX_LAUNCH_CHECK();
  // words of (row >> 8, sample): 32 bits when they fit, else 64
  int64_t max_card = 0;
  for (int f = 0; f < a->n_fields; ++f) {
    const int64_t card = a->field_base[f + 1] - a->field_base[f];
    max_card = card > max_card ? card : max_card;
  }
  const bool wide = pl.BB + (max_card > 256 ? fg_bits((uint64_t)((max_card - 1) >> 8)) : 0) > 32;
  const dim3 gw((unsigned)rx_cdiv((int64_t)pl.nf * 256, RX_BLOCK / 64));
  if (wide) {
    hipLaunchKernelGGL(fg_scatter_kernel<true>, dim3(pl.NT, pl.nf), dim3(RX_BLOCK), 0, st, pl, w.keyT, w.counts, w.part, w.bstart, w.biglist, d_srow, d_sb);
    RX_LAUNCH_CHECK();
    hipLaunchKernelGGL(fg_bin_wave_kernel<true>, gw, dim3(RX_BLOCK), 0, st, pl, w.part, w.bstart, d_srow, d_sb);
    RX_LAUNCH_CHECK();
    hipLaunchKernelGGL(fg_bin_big_kernel<true>, dim3(FG_BIG_GRID, pl.nf), dim3(1024), 0, st, pl, w.part, w.spare, w.bstart, w.biglist, d_srow, d_sb);
  } else {
    hipLaunchKernelGGL(fg_scatter_kernel<false>, dim3(pl.NT, pl.nf), dim3(RX_BLOCK), 0, st, pl, w.keyT, w.counts, w.part, w.bstart, w.biglist, d_srow, d_sb);
    RX_LAUNCH_CHECK();
    hipLaunchKernelGGL(fg_bin_wave_kernel<false>, gw, dim3(RX_BLOCK), 0, st, pl, w.part, w.bstart, d_srow, d_sb);
    RX_LAUNCH_CHECK();
    hipLaunchKernelGGL(fg_bin_big_kernel<false>, dim3(FG_BIG_GRID, pl.nf), dim3(1024), 0, st, pl, w.part, w.spare, w.bstart, w.biglist, d_srow, d_sb);
  }
  RX_LAUNCH_CHECK();
  return 0;
}

// ================================================================= reduce over the grouped list
#define FG_NONE 0xFFFFFFFFu       // row of a position past the end of the list
#define FG_NO_LEFT 0xFFFFFFFEu    // "row" left of position 0 / no carry / no tail

struct FgRed {
  const float* T; int64_t ldT;
  const float* g; int64_t ldg; int premul;   // g[b * ldg]; premul: T already holds g_b * T_b
  const float* deep; int64_t deep_ld;
  const uint32_t* srow; const uint32_t* sb;
  int64_t n; int B; int nf; int dim;
  float* target; float* state; const float* table; float* target2; float* state2;
  int64_t n_rows; float lr, eps; const float* scale; float* sq_part;
  float* partials;     // [windows][2][dim]: [0] run that entered from the left, [1] run that leaves to the right
  float* partials2;    // [windows][2]
  uint32_t* tail_row;  // [windows] row of the run that starts in the window and leaves it, or FG_NO_LEFT
  uint32_t* meta;      // [windows] leading positions held by the run entering from the left (| FG_META_GOES_ON)
  // rx_fg_row_update_clip: the clip factor is formed by every block from the partial sums of squares
  const float* clip_parts; int64_t clip_n; float clip_max; float* clip_scale_out; float* clip_norm_out;
  int64_t ldw, ldw2;   // row strides of the table (a.table, or a.target in the update modes) and of the companion table
  int line;            // the companion entry sits right behind its row in a line of 2 * dim floats: SGD writes whole lines
  float* undo;         // RX_SEG_SGD_SPEC: [positions][2 * dim] = gradient | the row before the step, at each run's last position
  const double* loss_parts; int n_loss_parts; int64_t loss_count; double* loss_out;   // deferred final sum of the forward's loss partials
  int force;           // RX_SEG_SGD_SPEC update: redo the rows even for a clip factor of 1 (they were put back before)
  float* undo2;        // ... and [positions][2] = companion gradient | companion entry before the step (NULL in the line layout,
                       //     where the two ride in the row's own line behind the entry)
};

// Second half of a row's line in the whole-line layout: the companion entry, then zeros.
template <int GROUP>
__device__ __forceinline__ void fg_store_line_tail(float* w_row, float v2, int lg) {
  *reinterpret_cast<float4*>(w_row + GROUP * 4 + lg * 4) = make_float4(lg == 0 ? v2 : 0.f, 0.f, 0.f, 0.f);
}

// The same in the speculative step: entry, its gradient and its value before the step, then zeros.
template <int GROUP>
__device__ __forceinline__ void fg_store_line_tail_undo(float* w_row, float v2, float g2, float old2, int lg) {
  *reinterpret_cast<float4*>(w_row + GROUP * 4 + lg * 4) = lg == 0 ? make_float4(v2, g2, old2, 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
}

// Everything that happens once the sum of a run is known: G = sum_b g_b T_b (+ tower slices),
// g2 = sum_b g_b; e: the row's current value, whose FM term g2 * e is subtracted here.  p: the run's
// last position (index of its compact gradient); old2: the companion entry's current value (read by
// the caller ahead of time; SGD only).  Called by all GROUP lanes of the run's lane group.
template <int GROUP, int MODE>
__device__ __forceinline__ float fg_finish(const FgRed& a, float4 G, float g2, float4 e, float old2, uint32_t row, int64_t p, int lg, float scale) {
  const int dim = GROUP * 4;
  const bool has2 = a.target2 != nullptr;
  G.x -= g2 * e.x; G.y -= g2 * e.y; G.z -= g2 * e.z; G.w -= g2 * e.w;
  float sq = 0.f;
  if (a.sq_part) {
    sq = G.x * G.x + G.y * G.y + G.z * G.z + G.w * G.w;
    if (has2 && lg == 0) sq += g2 * g2;
  }
  if constexpr (MODE == RX_SEG_COMPACT || MODE == RX_SEG_DENSE) {
    const int64_t at = MODE == RX_SEG_COMPACT ? p : (int64_t)row;
    *reinterpret_cast<float4*>(a.target + at * dim + lg * 4) = G;
    if (has2 && lg == 0) a.target2[at] = g2;
    return sq;
  }
  float4 x = e;
  float* w = a.target + (int64_t)row * a.ldw + lg * 4;
  if constexpr (MODE == RX_SEG_SGD_SPEC) {
    // The step as if the clip factor were 1 (times one is exact, so these are the bits of the clipped
    // step whenever the norm stays inside the bound), plus what the second launch needs to redo the
    // row exactly when it does not: the gradient and the row as it was.
    // (records are written once and, unless the clip bites, never read: streaming stores keep them from
    // displacing table lines in the cache -- 68 -> 60 us for this launch at B = 65536 x 26 fields)
    float* u = a.undo + p * (2 * dim);
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f Gv = {G.x, G.y, G.z, G.w}, ev = {e.x, e.y, e.z, e.w};
    __builtin_nontemporal_store(Gv, reinterpret_cast<v4f*>(u + lg * 4));
    __builtin_nontemporal_store(ev, reinterpret_cast<v4f*>(u + dim + lg * 4));
    x.x = x.x - a.lr * G.x; x.y = x.y - a.lr * G.y; x.z = x.z - a.lr * G.z; x.w = x.w - a.lr * G.w;
    const float new2 = old2 - a.lr * g2;
    if (a.line) fg_store_line_tail_undo<GROUP>(a.target + (int64_t)row * a.ldw, new2, g2, old2, lg);
    else if (has2 && lg == 0) {
      a.target2[(int64_t)row * a.ldw2] = new2;
      *reinterpret_cast<float2*>(a.undo2 + p * 2) = make_float2(g2, old2);
    }
    *reinterpret_cast<float4*>(w) = x;
    return sq;
  }
  G.x *= scale; G.y *= scale; G.z *= scale; G.w *= scale;
  if constexpr (MODE == RX_SEG_SGD) {
    x.x = x.x - a.lr * G.x; x.y = x.y - a.lr * G.y; x.z = x.z - a.lr * G.z; x.w = x.w - a.lr * G.w;
    const float new2 = old2 - a.lr * (g2 * scale);
    if (a.line) fg_store_line_tail<GROUP>(a.target + (int64_t)row * a.ldw, new2, lg);   // lane 0 of the group holds the entry
    else if (has2 && lg == 0) a.target2[(int64_t)row * a.ldw2] = new2;
  } else if constexpr (MODE == RX_SEG_ADAGRAD) {
    float* sp = a.state + (int64_t)row * dim + lg * 4;
    float4 s = *reinterpret_cast<const float4*>(sp);
    s.x += G.x * G.x; s.y += G.y * G.y; s.z += G.z * G.z; s.w += G.w * G.w;
    x.x -= a.lr * (G.x / (sqrtf(s.x) + a.eps)); x.y -= a.lr * (G.y / (sqrtf(s.y) + a.eps));
    x.z -= a.lr * (G.z / (sqrtf(s.z) + a.eps)); x.w -= a.lr * (G.w / (sqrtf(s.w) + a.eps));
    *reinterpret_cast<float4*>(sp) = s;
  } else {  // row-wise Adagrad: one state float per row
    float q = G.x * G.x + G.y * G.y + G.z * G.z + G.w * G.w;
    q = rx_group_sum(q, GROUP);
    const float s_new = a.state[row] + q / (float)dim;
    const float step = a.lr / (sqrtf(s_new) + a.eps);
    x.x -= step * G.x; x.y -= step * G.y; x.z -= step * G.z; x.w -= step * G.w;
    if (lg == 0) a.state[row] = s_new;
  }
  if constexpr (MODE == RX_SEG_ADAGRAD || MODE == RX_SEG_ROWWISE_ADAGRAD) {
    if (has2 && lg == 0) {  // the [rows, 1] companion: Adagrad on a width-1 row
      const float gs = g2 * scale;
      const float s2 = a.state2[row] + gs * gs;
      a.state2[row] = s2;
      a.target2[(int64_t)row * a.ldw2] = a.target2[(int64_t)row * a.ldw2] - a.lr * (gs / (sqrtf(s2) + a.eps));
    }
  }
  *reinterpret_cast<float4*>(w) = x;
  return sq;
}

__device__ __forceinline__ void fg_block_sq_store(float sq, float* sq_part, float* sq_lds) {
  sq = rx_wave_sum(sq);
  if ((threadIdx.x & 63) == 0) sq_lds[threadIdx.x >> 6] = sq;
  __syncthreads();
  if (threadIdx.x == 0) sq_part[blockIdx.x] = (sq_lds[0] + sq_lds[1]) + (sq_lds[2] + sq_lds[3]);
}

// One wavefront per window of W = S * (64 / GROUP) consecutive positions.  A lane group of GROUP
// lanes (16 bytes each) owns S CONSECUTIVE positions: it loads their rows, samples, T rows and — for
// the positions that end a run — the run's current table row up front, then walks the S positions
// in registers.  Runs that begin and end inside the group are finished during the walk.  What
// crosses group boundaries is joined by one segmented scan over the groups' trailing sums (fixed
// tree), what crosses the window goes to `partials` for fg_combine_kernel.
// meta[window] describes the run that enters from the left: the number of leading positions it
// holds in this window, bit 31 set when it goes on past the window's end.
#define FG_META_GOES_ON 0x80000000u
template <int GROUP, int S, int MODE, bool DEEP>
__global__ void __launch_bounds__(RX_BLOCK)
fg_reduce_kernel(FgRed a) {
  static_assert(S == 8, "the row loads below are written for 8 positions per lane group");
  constexpr int GPW = 64 / GROUP;
  constexpr int W = GPW * S;
  constexpr int DIM = GROUP * 4;
  __shared__ float sq_lds[RX_BLOCK / 64];
  const int lane = threadIdx.x & 63;
  const int grp = lane / GROUP, lg = lane % GROUP;
  const int64_t window = (int64_t)blockIdx.x * (RX_BLOCK / 64) + (threadIdx.x >> 6);
  const int64_t wbase = window * W;
  float sq = 0.f;
  if (wbase < a.n) {
    const float scale = a.scale ? a.scale[0] : 1.f;
    const float* tab = (MODE >= RX_SEG_SGD) ? a.target : a.table;
    const bool has2 = a.target2 != nullptr;
    const int64_t gbase = wbase + grp * S;
    // rows of positions gbase - 1 .. gbase + S, samples of gbase .. gbase + S - 1
    uint32_t rw[S + 2], bb[S];
    if (gbase + S < a.n) {
      const uint4 r0 = *reinterpret_cast<const uint4*>(a.srow + gbase), r1 = *reinterpret_cast<const uint4*>(a.srow + gbase + 4);
      const uint4 b0 = *reinterpret_cast<const uint4*>(a.sb + gbase), b1 = *reinterpret_cast<const uint4*>(a.sb + gbase + 4);
      rw[1] = r0.x; rw[2] = r0.y; rw[3] = r0.z; rw[4] = r0.w; rw[5] = r1.x; rw[6] = r1.y; rw[7] = r1.z; rw[8] = r1.w;
      bb[0] = b0.x; bb[1] = b0.y; bb[2] = b0.z; bb[3] = b0.w; bb[4] = b1.x; bb[5] = b1.y; bb[6] = b1.z; bb[7] = b1.w;
      rw[S + 1] = a.srow[gbase + S];
    } else {
#pragma unroll
      for (int s = 0; s <= S; ++s) {
        const int64_t p = gbase + s;
        rw[s + 1] = p < a.n ? a.srow[p] : FG_NONE;
        if (s < S) bb[s] = p < a.n ? a.sb[p] : 0u;
      }
    }
    rw[0] = gbase > 0 ? (gbase - 1 < a.n ? a.srow[gbase - 1] : FG_NONE) : FG_NO_LEFT;
    bool endf[S];
    float4 v[S], e[S];
    float w2[S], old2[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
      // a position ends a run when the next row differs; the run's own row (FM term, update) is
      // fetched for those positions only
      endf[s] = rw[s + 1] != FG_NONE && rw[s + 2] != rw[s + 1] && (int64_t)rw[s + 1] < a.n_rows;
      e[s] = make_float4(0.f, 0.f, 0.f, 0.f);
      old2[s] = 0.f;
      if (endf[s]) {
        e[s] = *reinterpret_cast<const float4*>(tab + (int64_t)rw[s + 1] * a.ldw + lg * 4);
        if ((MODE == RX_SEG_SGD || MODE == RX_SEG_SGD_SPEC) && has2 && lg == 0) old2[s] = a.target2[(int64_t)rw[s + 1] * a.ldw2];
      }
    }
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const bool valid = rw[s + 1] != FG_NONE;
      const float gb = a.g[(int64_t)bb[s] * a.ldg];
      const float4 t = *reinterpret_cast<const float4*>(a.T + (int64_t)bb[s] * a.ldT + lg * 4);
      const float gv = valid ? gb : 0.f;
      const float mv = valid ? (a.premul ? 1.f : gb) : 0.f;   // times one is exact: premultiplied rows pass through
      v[s] = make_float4(mv * t.x, mv * t.y, mv * t.z, mv * t.w);
      w2[s] = gv;
      if constexpr (DEEP) {
        // the occurrence's slice of the tower-input gradient: [last field | field 0 .. nf-2 | ...]
        const int f = (int)((valid ? gbase + s : 0) / a.B);
        const int slice = f == a.nf - 1 ? 0 : 1 + f;
        const float4 dd = *reinterpret_cast<const float4*>(a.deep + (int64_t)bb[s] * a.deep_ld + slice * DIM + lg * 4);
        if (valid) { v[s].x += dd.x; v[s].y += dd.y; v[s].z += dd.z; v[s].w += dd.w; }
      }
    }
    // the walk: position order inside the group
    const bool gvalid = gbase < a.n;
    const bool cont_in = gvalid && rw[1] == rw[0];   // the first position's run began before this group
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), hsum = acc, he = acc;
    float acc2 = 0.f, h2 = 0.f, hold2 = 0.f;
    int hs = -1;
    uint32_t hrow = 0;
    bool seen_end = false;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      acc.x += v[s].x; acc.y += v[s].y; acc.z += v[s].z; acc.w += v[s].w;
      acc2 += w2[s];
      if (endf[s]) {
        if (cont_in && !seen_end) {  // needs what the groups before this one hold: kept for after the scan
          hsum = acc; h2 = acc2; he = e[s]; hold2 = old2[s]; hs = s; hrow = rw[s + 1];
        } else {
          sq += fg_finish<GROUP, MODE>(a, acc, acc2, e[s], old2[s], rw[s + 1], gbase + s, lg, scale);
        }
        seen_end = true;
        acc = make_float4(0.f, 0.f, 0.f, 0.f);
        acc2 = 0.f;
      }
    }
    // segmented inclusive scan of the trailing sums over the groups: a group is "open" when it is one
    // run from end to end that also began before it, so that its sum joins its left neighbour's
    float4 X = acc;
    float X2 = acc2;
    int open = gvalid && !seen_end && cont_in;
#pragma unroll
    for (int off = 1; off < GPW; off <<= 1) {
      const float tx = __shfl_up(X.x, off * GROUP, 64), ty = __shfl_up(X.y, off * GROUP, 64);
      const float tz = __shfl_up(X.z, off * GROUP, 64), tw = __shfl_up(X.w, off * GROUP, 64);
      const float t2 = __shfl_up(X2, off * GROUP, 64);
      const int to = __shfl_up(open, off * GROUP, 64);
      if (grp >= off && open) {
        X.x = tx + X.x; X.y = ty + X.y; X.z = tz + X.z; X.w = tw + X.w;
        X2 = t2 + X2;
        open = to;
      }
    }
    // what the left neighbour hands over, and whether that chain reaches back past the window's start
    float4 C = make_float4(__shfl_up(X.x, GROUP, 64), __shfl_up(X.y, GROUP, 64), __shfl_up(X.z, GROUP, 64), __shfl_up(X.w, GROUP, 64));
    float C2 = __shfl_up(X2, GROUP, 64);
    int from_left = __shfl_up(open, GROUP, 64);
    if (grp == 0) { C = make_float4(0.f, 0.f, 0.f, 0.f); C2 = 0.f; from_left = 1; }
    float* part0 = a.partials + window * 2 * DIM;
    float* part0_2 = a.partials2 + window * 2;
    if (lane == 0 && !cont_in) a.meta[window] = 0u;   // no run enters the window from the left
    if (hs >= 0) {
      const float4 G = make_float4(C.x + hsum.x, C.y + hsum.y, C.z + hsum.z, C.w + hsum.w);
      const float g2 = C2 + h2;
      if (from_left) {  // the run came in from the left neighbour window: its sum over this window's part
        *reinterpret_cast<float4*>(part0 + lg * 4) = G;
        if (lg == 0) {
          part0_2[0] = g2;
          a.meta[window] = (uint32_t)(grp * S + hs + 1);
        }
      } else {
        sq += fg_finish<GROUP, MODE>(a, G, g2, he, hold2, hrow, gbase + hs, lg, scale);
      }
    }
    // the run that leaves the window to the right (only a full window can have one)
    if (grp == GPW - 1) {
      uint32_t tail = FG_NO_LEFT;
      if (rw[S] != FG_NONE && !endf[S - 1] && (int64_t)rw[S] < a.n_rows) {
        const int side = open ? 0 : 1;   // open after the scan: one run from the window's start, begun before it
        *reinterpret_cast<float4*>(part0 + side * DIM + lg * 4) = X;
        if (lg == 0) part0_2[side] = X2;
        if (side == 1) tail = rw[S];
        else if (lg == 0) a.meta[window] = (uint32_t)W | FG_META_GOES_ON;
      }
      if (lg == 0) a.tail_row[window] = tail;
    }
  }
  if (a.loss_out && blockIdx.x == gridDim.x - 1 && threadIdx.x < 64) {
    // the forward left its per-block fp64 loss partials (rx_fm2_fwd_bce with d_loss == NULL): one wavefront of
    // the last block adds them exactly as bce_final_kernel does (lane l takes l, l + 64, ...; fixed xor tree)
    double s = 0.0;
    for (int i = threadIdx.x; i < a.n_loss_parts; i += 64) s += a.loss_parts[i];
    s = rx_wave_sum_f64(s);
    if (threadIdx.x == 0) a.loss_out[0] = s / (double)a.loss_count;
  }
  if (a.sq_part) fg_block_sq_store(sq, a.sq_part, sq_lds);
}

// Joins the partial sums of runs that cross window boundaries.  A lane group looks after one
// window: when the run that starts there ends in the next window (the common case) the group adds
// the two partials and finishes the row on its own.  Runs that span more windows (hot rows) are
// taken one by one by the whole wavefront: its lane slots add the run's partials (slot j takes
// windows j, j + slots, ... in order), a fixed xor tree joins the slots.
template <int GROUP, int S, int MODE>
__global__ void __launch_bounds__(RX_BLOCK)
fg_combine_kernel(FgRed a, int64_t n_windows) {
  constexpr int GPW = 64 / GROUP;
  constexpr int W = GPW * S;
  constexpr int DIM = GROUP * 4;
  __shared__ float sq_lds[RX_BLOCK / 64];
  const int lane = threadIdx.x & 63;
  const int slot = lane / GROUP, lg = lane % GROUP;
  const int64_t w0 = ((int64_t)blockIdx.x * (RX_BLOCK / 64) + (threadIdx.x >> 6)) * GPW;
  const float scale = a.scale ? a.scale[0] : 1.f;
  const float* tab = (MODE >= RX_SEG_SGD) ? a.target : a.table;
  float sq = 0.f;
  const int64_t win = w0 + slot;
  const bool in = win < n_windows;
  // everything the two-window case needs, loaded before the tail is known
  const uint32_t tail = in ? a.tail_row[win] : FG_NO_LEFT;
  const bool nxt_in = win + 1 < n_windows;
  const uint32_t meta1 = nxt_in ? a.meta[win + 1] : 0u;
  float4 G = make_float4(0.f, 0.f, 0.f, 0.f), H = G;
  float g2a = 0.f, g2b = 0.f;
  if (in) {
    G = *reinterpret_cast<const float4*>(a.partials + (win * 2 + 1) * DIM + lg * 4);
    g2a = a.partials2[win * 2 + 1];
  }
  if (nxt_in) {
    H = *reinterpret_cast<const float4*>(a.partials + (win + 1) * 2 * DIM + lg * 4);
    g2b = a.partials2[(win + 1) * 2];
  }
  const bool has_tail = tail != FG_NO_LEFT;
  const bool shortrun = has_tail && !(meta1 & FG_META_GOES_ON);
  if (shortrun) {
    const float4 e = *reinterpret_cast<const float4*>(tab + (int64_t)tail * a.ldw + lg * 4);
    const float old2 = ((MODE == RX_SEG_SGD || MODE == RX_SEG_SGD_SPEC) && a.target2 && lg == 0) ? a.target2[(int64_t)tail * a.ldw2] : 0.f;
    G.x += H.x; G.y += H.y; G.z += H.z; G.w += H.w;
    const int64_t p_end = (win + 1) * W + (int64_t)(meta1 & 0xFFFFu) - 1;
    sq += fg_finish<GROUP, MODE>(a, G, g2a + g2b, e, old2, tail, p_end, lg, scale);
  }
  unsigned long long todo = __ballot(has_tail && !shortrun && lg == 0);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;  // wave-uniform
    todo &= todo - 1;
    const uint32_t row = (uint32_t)__shfl((int)tail, leader, 64);
    const int64_t wl = w0 + leader / GROUP;
    // windows wl+1 .. wl+cnt hold the run's continuation; the last of them is the first whose meta
    // does not say "goes on"
    int64_t cnt = 0;
    uint32_t last_meta = 0;
    for (;;) {
      const int64_t wq = wl + 1 + cnt + lane;
      const uint32_t mq = wq < n_windows ? a.meta[wq] : 0u;
      const unsigned long long m = __ballot((mq & FG_META_GOES_ON) != 0);
      const int run = m == ~0ull ? 64 : __ffsll((long long)~m) - 1;
      if (run < 64) {
        last_meta = (uint32_t)__shfl((int)mq, run, 64);
        cnt += run + 1;
        break;
      }
      cnt += 64;
    }
    const int64_t p_end = (wl + cnt) * W + (int64_t)(last_meta & 0xFFFFu) - 1;
    float4 A = make_float4(0.f, 0.f, 0.f, 0.f);
    float a2 = 0.f;
    for (int64_t q = slot; q <= cnt; q += GPW) {
      const int64_t ee = (wl + q) * 2 + (q == 0 ? 1 : 0);
      const float4 t = *reinterpret_cast<const float4*>(a.partials + ee * DIM + lg * 4);
      A.x += t.x; A.y += t.y; A.z += t.z; A.w += t.w;
      a2 += a.partials2[ee];
    }
#pragma unroll
    for (int m = GROUP; m < 64; m <<= 1) {
      A.x += __shfl_xor(A.x, m, 64); A.y += __shfl_xor(A.y, m, 64);
      A.z += __shfl_xor(A.z, m, 64); A.w += __shfl_xor(A.w, m, 64);
      a2 += __shfl_xor(a2, m, 64);
    }
    if (slot == 0) {
      const float4 e = *reinterpret_cast<const float4*>(tab + (int64_t)row * a.ldw + lg * 4);
      const float old2 = ((MODE == RX_SEG_SGD || MODE == RX_SEG_SGD_SPEC) && a.target2 && lg == 0) ? a.target2[(int64_t)row * a.ldw2] : 0.f;
      sq += fg_finish<GROUP, MODE>(a, A, a2, e, old2, row, p_end, lg, scale);
    }
  }
  if (a.sq_part) fg_block_sq_store(sq, a.sq_part, sq_lds);
}

// Second phase of a clipped step: rows move by their compact gradients (stored at the last
// position of each run) times the clip factor.  A lane group looks at FG_UPD consecutive positions.
#define FG_UPD 8   // the row loads of the update kernel are written for 8
// The global-norm clip factor from per-block partial sums of squares, formed by EVERY block of the
// consuming launch instead of by a one-block launch in front of it (4.6 us on the step's critical
// path for a few thousand floats that sit in L2).  Same arithmetic and order as
// clip_scale_parts_kernel: thread t adds parts[t], parts[t + 256], ... in fp64, a fixed tree joins
// the threads; all blocks therefore hold the same bits.
__device__ __forceinline__ float fg_clip_scale(const FgRed& a, double* wsum) {
  double s = 0.0;
  for (int64_t i0 = 0; i0 < a.clip_n; i0 += 8 * RX_BLOCK) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {   // loads first (entries past the end read as 0 and leave the sum's bits alone)
      const int64_t i = i0 + u * RX_BLOCK + threadIdx.x;
      v[u] = a.clip_parts[i < a.clip_n ? i : 0];
      if (i >= a.clip_n) v[u] = 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) s += (double)v[u];
  }
  s = rx_wave_sum_f64(s);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
  __syncthreads();
  const double total = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
  const float tn = (float)sqrt(total);
  const float coef = a.clip_max / (tn + 1e-6f);
  const float scale = coef < 1.f ? coef : 1.f;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    a.clip_scale_out[0] = scale;
    if (a.clip_norm_out) a.clip_norm_out[0] = tn;
  }
  return scale;
}

template <int GROUP, int MODE, bool CLIP = false>
__global__ void __launch_bounds__(RX_BLOCK)
fg_row_update_kernel(FgRed a, const float* __restrict__ compact, const float* __restrict__ compact2) {
  constexpr int DIM = GROUP * 4;
  __shared__ double clip_wsum[RX_BLOCK / 64];
  float scale = 1.f;
  if constexpr (CLIP) scale = fg_clip_scale(a, clip_wsum);   // before any thread leaves: it holds a block barrier
  else scale = a.scale ? a.scale[0] : 1.f;
  if constexpr (MODE == RX_SEG_SGD_SPEC) {
    // the reduce already took the step with a factor of 1: nothing to do unless the clip bites
    if (scale == 1.f && !a.force) return;
  }
  const int lg = threadIdx.x % GROUP;
  // (the speculative step launches a small grid that strides over the list: in the common case every block
  // only forms the clip factor and ends, so few blocks make that cheap; the other modes launch one lane
  // group per FG_UPD positions and take the loop once)
  for (int64_t gid = ((int64_t)blockIdx.x * RX_BLOCK + threadIdx.x) / GROUP; gid * FG_UPD < a.n;
       gid += (int64_t)gridDim.x * (RX_BLOCK / GROUP)) {
  const int64_t p0 = gid * FG_UPD;
  const bool has2 = a.target2 != nullptr;
  uint32_t row[FG_UPD + 1];
  if (p0 + FG_UPD < a.n) {
    const uint4 r0 = *reinterpret_cast<const uint4*>(a.srow + p0), r1 = *reinterpret_cast<const uint4*>(a.srow + p0 + 4);
    row[0] = r0.x; row[1] = r0.y; row[2] = r0.z; row[3] = r0.w; row[4] = r1.x; row[5] = r1.y; row[6] = r1.z; row[7] = r1.w;
    row[8] = a.srow[p0 + 8];
  } else {
#pragma unroll
    for (int j = 0; j <= FG_UPD; ++j) row[j] = p0 + j < a.n ? a.srow[p0 + j] : FG_NONE;
  }
  bool endf[FG_UPD];
  float4 G[FG_UPD], x[FG_UPD];
  float g2[FG_UPD], old2[FG_UPD];
#pragma unroll
  for (int j = 0; j < FG_UPD; ++j) {
    endf[j] = row[j] != FG_NONE && row[j] != row[j + 1] && (int64_t)row[j] < a.n_rows;
    g2[j] = 0.f; old2[j] = 0.f;
    if (endf[j]) {
      if constexpr (MODE == RX_SEG_SGD_SPEC) {   // gradient and the row before the step, side by side
        G[j] = *reinterpret_cast<const float4*>(compact + (p0 + j) * (2 * DIM) + lg * 4);
        x[j] = *reinterpret_cast<const float4*>(compact + (p0 + j) * (2 * DIM) + DIM + lg * 4);
        if (has2 && lg == 0) {
          if (a.line) {
            const float4 t = *reinterpret_cast<const float4*>(a.target + (int64_t)row[j] * a.ldw + DIM);
            g2[j] = t.y; old2[j] = t.z;
          } else {
            const float2 t = *reinterpret_cast<const float2*>(compact2 + (p0 + j) * 2);
            g2[j] = t.x; old2[j] = t.y;
          }
        }
      } else {
        G[j] = *reinterpret_cast<const float4*>(compact + (p0 + j) * DIM + lg * 4);
        x[j] = *reinterpret_cast<const float4*>(a.target + (int64_t)row[j] * a.ldw + lg * 4);
        if (has2 && lg == 0) {
          g2[j] = compact2[p0 + j];
          if (MODE == RX_SEG_SGD) old2[j] = a.target2[(int64_t)row[j] * a.ldw2];
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < FG_UPD; ++j) {
    if (!endf[j]) continue;
    const uint32_t r = row[j];
    float4 g = G[j], w = x[j];
    g.x *= scale; g.y *= scale; g.z *= scale; g.w *= scale;
    if constexpr (MODE == RX_SEG_SGD || MODE == RX_SEG_SGD_SPEC) {
      w.x = w.x - a.lr * g.x; w.y = w.y - a.lr * g.y; w.z = w.z - a.lr * g.z; w.w = w.w - a.lr * g.w;
      const float new2 = old2[j] - a.lr * (g2[j] * scale);
      if (a.line) {
        if constexpr (MODE == RX_SEG_SGD_SPEC) fg_store_line_tail_undo<GROUP>(a.target + (int64_t)r * a.ldw, new2, g2[j], old2[j], lg);
        else fg_store_line_tail<GROUP>(a.target + (int64_t)r * a.ldw, new2, lg);   // lane 0 of the group holds the entry
      } else if (has2 && lg == 0) a.target2[(int64_t)r * a.ldw2] = new2;
    } else if constexpr (MODE == RX_SEG_ADAGRAD) {
      float* sp = a.state + (int64_t)r * DIM + lg * 4;
      float4 s = *reinterpret_cast<const float4*>(sp);
      s.x += g.x * g.x; s.y += g.y * g.y; s.z += g.z * g.z; s.w += g.w * g.w;
      w.x -= a.lr * (g.x / (sqrtf(s.x) + a.eps)); w.y -= a.lr * (g.y / (sqrtf(s.y) + a.eps));
      w.z -= a.lr * (g.z / (sqrtf(s.z) + a.eps)); w.w -= a.lr * (g.w / (sqrtf(s.w) + a.eps));
      *reinterpret_cast<float4*>(sp) = s;
    } else if constexpr (MODE == RX_SEG_ROWWISE_ADAGRAD) {
      float q = g.x * g.x + g.y * g.y + g.z * g.z + g.w * g.w;
      q = rx_group_sum(q, GROUP);
      const float s_new = a.state[r] + q / (float)DIM;
      const float step = a.lr / (sqrtf(s_new) + a.eps);
      w.x -= step * g.x; w.y -= step * g.y; w.z -= step * g.z; w.w -= step * g.w;
      if (lg == 0) a.state[r] = s_new;
    }
    if constexpr (MODE == RX_SEG_ADAGRAD || MODE == RX_SEG_ROWWISE_ADAGRAD) {
      if (has2 && lg == 0) {
        const float gs = g2[j] * scale;
        const float s2 = a.state2[r] + gs * gs;
        a.state2[r] = s2;
        a.target2[(int64_t)r * a.ldw2] = a.target2[(int64_t)r * a.ldw2] - a.lr * (gs / (sqrtf(s2) + a.eps));
      }
    }
    *reinterpret_cast<float4*>(a.target + (int64_t)r * a.ldw + lg * 4) = w;
  }
  }
}

// 8 steps per window: 128 / 64 / 32 / 16 positions for 16 / 32 / 64 / 128 columns
static int fg_group_of(int dim) {
  switch (dim) { case 16: return 4; case 32: return 8; case 64: return 16; case 128: return 32; default: return 0; }
}
static int64_t fg_window(int dim) { return (int64_t)(64 / fg_group_of(dim)) * 8; }
static int64_t fg_windows(int64_t n, int dim) { return rx_cdiv(n, fg_window(dim)); }
static int64_t fg_grid1(int64_t n, int dim) { return rx_cdiv(fg_windows(n, dim), RX_BLOCK / 64); }
static int64_t fg_grid2(int64_t n, int dim) { return rx_cdiv(fg_windows(n, dim), (RX_BLOCK / 64) * (64 / fg_group_of(dim))); }

extern "C" int rx_fg_dim_supported(int32_t dim) { return fg_group_of(dim) != 0; }

extern "C" int64_t rx_fg_reduce_grid(int64_t n, int32_t dim) {
  if (n <= 0 || !fg_group_of(dim)) return 0;
  return fg_grid1(n, dim) + fg_grid2(n, dim);
}

extern "C" size_t rx_fg_reduce_workspace_bytes(int64_t n, int32_t dim) {
  if (n <= 0 || !fg_group_of(dim)) return 256;
  const size_t w = (size_t)fg_windows(n, dim);
  return rx_align_up(w * 2 * dim * sizeof(float), 256) + rx_align_up(w * 2 * sizeof(float), 256) + 2 * rx_align_up(w * sizeof(uint32_t), 256);
}

#define FG_LAUNCH_MODE(GROUPV, MODEV)                                                                          \
  do {                                                                                                         \
    if (deep) hipLaunchKernelGGL((fg_reduce_kernel<GROUPV, 8, MODEV, true>), dim3(g1), dim3(RX_BLOCK), 0, st, a); \
    else hipLaunchKernelGGL((fg_reduce_kernel<GROUPV, 8, MODEV, false>), dim3(g1), dim3(RX_BLOCK), 0, st, a);  \
    a2 = a;                                                                                                    \
    a2.loss_out = nullptr;                                                                                     \
    if (a2.sq_part) a2.sq_part += g1;                                                                          \
    hipLaunchKernelGGL((fg_combine_kernel<GROUPV, 8, MODEV>), dim3(g2), dim3(RX_BLOCK), 0, st, a2, n_windows); \
  } while (0)

#define FG_LAUNCH_GROUP(GROUPV)                                                        \
  do {                                                                                 \
    switch (mode) {                                                                    \
      case RX_SEG_DENSE: FG_LAUNCH_MODE(GROUPV, RX_SEG_DENSE); break;                  \
      case RX_SEG_COMPACT: FG_LAUNCH_MODE(GROUPV, RX_SEG_COMPACT); break;              \
      case RX_SEG_SGD: FG_LAUNCH_MODE(GROUPV, RX_SEG_SGD); break;                      \
      case RX_SEG_ADAGRAD: FG_LAUNCH_MODE(GROUPV, RX_SEG_ADAGRAD); break;              \
      case RX_SEG_SGD_SPEC: FG_LAUNCH_MODE(GROUPV, RX_SEG_SGD_SPEC); break;            \
      default: FG_LAUNCH_MODE(GROUPV, RX_SEG_ROWWISE_ADAGRAD); break;                  \
    }                                                                                  \
  } while (0)

extern "C" int rx_fg_reduce(const rx_fg_reduce_args* r, void* ws, size_t ws_bytes, int device, void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(r, "rx_fg_reduce: null args");
  const int64_t B = r->B;
  const int n_fields = r->n_fields, dim = r->dim, mode = r->mode;
  RX_REQUIRE(B >= 0 && n_fields >= 1 && n_fields <= RX_FG_MAX_FIELDS, "rx_fg_reduce: bad shape");
  RX_REQUIRE(fg_group_of(dim), "rx_fg_reduce: dim %d unsupported (16, 32, 64 or 128)", dim);
  RX_REQUIRE(mode >= RX_SEG_DENSE && mode <= RX_SEG_SGD_SPEC, "rx_fg_reduce: unknown mode %d", mode);
  const int64_t n = B * n_fields;
  if (n == 0) return 0;
  RX_REQUIRE(n < (1ll << 31), "rx_fg_reduce: too many occurrences");
  RX_REQUIRE(r->T && r->g && r->srow && r->sb && r->target && ws, "rx_fg_reduce: null argument");
  RX_REQUIRE(rx_aligned16(r->srow) && rx_aligned16(r->sb), "rx_fg_reduce: the grouped arrays must be 16-byte aligned");
  RX_REQUIRE(r->ld_T >= dim && r->ld_T % 4 == 0 && rx_aligned16(r->T) && rx_aligned16(r->target), "rx_fg_reduce: T / target must be 16-byte aligned with a row stride that is a multiple of 4");
  RX_REQUIRE(r->ld_g >= 1, "rx_fg_reduce: ld_g must be positive");
  const bool grad_mode = mode == RX_SEG_DENSE || mode == RX_SEG_COMPACT;
  RX_REQUIRE(!grad_mode || (r->table && rx_aligned16(r->table)), "rx_fg_reduce: gradient modes need the (16-byte aligned) table to read rows from");
  RX_REQUIRE(!(mode == RX_SEG_ADAGRAD || mode == RX_SEG_ROWWISE_ADAGRAD) || (r->state && (!r->target2 || r->state2)), "rx_fg_reduce: Adagrad needs its state buffers");
  RX_REQUIRE(mode != RX_SEG_ADAGRAD || rx_aligned16(r->state), "rx_fg_reduce: state must be 16-byte aligned");
  RX_REQUIRE(r->n_rows >= 1 && r->n_rows <= (1ll << 32) - 16, "rx_fg_reduce: n_rows out of range");
  RX_REQUIRE(ws_bytes >= rx_fg_reduce_workspace_bytes(n, dim) && rx_aligned16(ws), "rx_fg_reduce: workspace too small or unaligned");
  const bool deep = r->deep != nullptr;
  if (deep) RX_REQUIRE(r->deep_ld >= (int64_t)n_fields * dim && r->deep_ld % 4 == 0 && rx_aligned16(r->deep), "rx_fg_reduce: tower gradient must be 16-byte aligned, row stride a multiple of 4 and >= n_fields * dim");
  FgRed a{};
  a.T = r->T; a.ldT = r->ld_T; a.g = r->g; a.ldg = r->ld_g; a.premul = r->premultiplied; a.deep = r->deep; a.deep_ld = r->deep_ld;
  a.srow = r->srow; a.sb = r->sb;
  a.n = n; a.B = (int)B; a.nf = n_fields; a.dim = dim; a.target = r->target; a.state = r->state; a.table = r->table;
  a.target2 = r->target2; a.state2 = r->state2; a.n_rows = r->n_rows; a.lr = r->lr; a.eps = r->eps; a.scale = r->scale; a.sq_part = r->sq_part;
  a.ldw = r->ld_w ? r->ld_w : dim; a.ldw2 = r->ld_w2 ? r->ld_w2 : 1;
  RX_REQUIRE(a.ldw >= dim && a.ldw % 4 == 0 && a.ldw2 >= 1, "rx_fg_reduce: table row stride must be a multiple of 4 floats and >= dim");
  a.line = !grad_mode && (mode == RX_SEG_SGD || mode == RX_SEG_SGD_SPEC) && r->target2 && r->target2 == r->target + dim && a.ldw == 2 * dim && a.ldw2 == a.ldw;
  if (mode == RX_SEG_SGD_SPEC) {
    RX_REQUIRE(r->undo && rx_aligned16(r->undo), "rx_fg_reduce: the speculative step needs its (16-byte aligned) undo buffer [B * n_fields, 2 * dim]");
    RX_REQUIRE(!r->target2 || a.line || (r->undo2 && ((uintptr_t)r->undo2 & 7) == 0), "rx_fg_reduce: the speculative step needs undo2 [B * n_fields, 2] for a companion table outside the row's line");
    RX_REQUIRE(!a.line || dim >= 4, "rx_fg_reduce: line layout needs room for the undo pair");
    a.undo = r->undo; a.undo2 = a.line ? nullptr : r->undo2;
  }
  if (r->loss_out) {
    RX_REQUIRE(r->loss_parts && r->n_loss_parts >= 1 && r->loss_count >= 1, "rx_fg_reduce: loss_out needs the forward's partial sums, their count and the number of logits");
    a.loss_parts = r->loss_parts; a.n_loss_parts = r->n_loss_parts; a.loss_count = r->loss_count; a.loss_out = r->loss_out;
  }
  const size_t nw = (size_t)fg_windows(n, dim);
  char* w = (char*)ws;
  a.partials = (float*)w; w += rx_align_up(nw * 2 * dim * sizeof(float), 256);
  a.partials2 = (float*)w; w += rx_align_up(nw * 2 * sizeof(float), 256);
  a.tail_row = (uint32_t*)w; w += rx_align_up(nw * sizeof(uint32_t), 256);
  a.meta = (uint32_t*)w;
  const unsigned g1 = (unsigned)fg_grid1(n, dim), g2 = (unsigned)fg_grid2(n, dim);
  const int64_t n_windows = (int64_t)nw;
  hipStream_t st = (hipStream_t)stream;
  FgRed a2;
  switch (fg_group_of(dim)) {
    case 4: FG_LAUNCH_GROUP(4); break;
    case 8: FG_LAUNCH_GROUP(8); break;
    case 16: FG_LAUNCH_GROUP(16); break;
    default: FG_LAUNCH_GROUP(32); break;
  }
  RX_LAUNCH_CHECK();
  return 0;
}

#define FG_SPEC_GRID 512   /* blocks of the speculative step's second launch (two per CU) */
#define FG_UPDATE_GROUP(GROUPV)                                                                                                        \
  do {                                                                                                                                 \
    const unsigned grid = (unsigned)rx_cdiv(rx_cdiv(n, FG_UPD) * GROUPV, RX_BLOCK);                                                    \
    if (mode == RX_SEG_SGD_SPEC) {                                                                                                    \
      const unsigned sgrid = grid < FG_SPEC_GRID ? grid : FG_SPEC_GRID;                                                                \
      if (a.clip_parts) hipLaunchKernelGGL((fg_row_update_kernel<GROUPV, RX_SEG_SGD_SPEC, true>), dim3(sgrid), dim3(RX_BLOCK), 0, st, a, d_compact, d_compact2); \
      else hipLaunchKernelGGL((fg_row_update_kernel<GROUPV, RX_SEG_SGD_SPEC>), dim3(sgrid), dim3(RX_BLOCK), 0, st, a, d_compact, d_compact2); \
    } else if (a.clip_parts) {                                                                                                        \
      if (mode == RX_SEG_SGD) hipLaunchKernelGGL((fg_row_update_kernel<GROUPV, RX_SEG_SGD, true>), dim3(grid), dim3(RX_BLOCK), 0, st, a, d_compact, d_compact2); \
      else if (mode == RX_SEG_ADAGRAD) hipLaunchKernelGGL((fg_row_update_kernel<GROUPV, RX_SEG_ADAGRAD, true>), dim3(grid), dim3(RX_BLOCK), 0, st, a, d_compact, d_compact2); \
      else hipLaunchKernelGGL((fg_row_update_kernel<GROUPV, RX_SEG_ROWWISE_ADAGRAD, true>), dim3(grid), dim3(RX_BLOCK), 0, st, a, d_compact, d_compact2); \
    } else if (mode == RX_SEG_SGD) hipLaunchKernelGGL((fg_row_update_kernel<GROUPV, RX_SEG_SGD>), dim3(grid), dim3(RX_BLOCK), 0, st, a, d_compact, d_compact2); \
    else if (mode == RX_SEG_ADAGRAD) hipLaunchKernelGGL((fg_row_update_kernel<GROUPV, RX_SEG_ADAGRAD>), dim3(grid), dim3(RX_BLOCK), 0, st, a, d_compact, d_compact2); \
    else hipLaunchKernelGGL((fg_row_update_kernel<GROUPV, RX_SEG_ROWWISE_ADAGRAD>), dim3(grid), dim3(RX_BLOCK), 0, st, a, d_compact, d_compact2); \
  } while (0)

static int fg_row_update_run(const float* d_compact, int32_t dim, const uint32_t* d_srow, int64_t n, int mode,
                             float* d_w, float* d_state, const float* d_compact2, float* d_w2, float* d_state2,
                             int64_t n_rows, float lr, float eps, const float* d_scale, const float* d_parts,
                             int64_t n_parts, float max_norm, float* d_scale_out, float* d_norm_out, int64_t ld_w, int64_t ld_w2,
                             int device, void* stream);

extern "C" int rx_fg_row_update(const float* d_compact, int32_t dim, const uint32_t* d_srow, int64_t n, int mode,
                                float* d_w, float* d_state, const float* d_compact2, float* d_w2, float* d_state2,
                                int64_t n_rows, float lr, float eps, const float* d_scale, int64_t ld_w, int64_t ld_w2,
                                int device, void* stream) {
  return fg_row_update_run(d_compact, dim, d_srow, n, mode, d_w, d_state, d_compact2, d_w2, d_state2, n_rows, lr, eps, d_scale,
                           nullptr, 0, 0.f, nullptr, nullptr, ld_w, ld_w2, device, stream);
}

extern "C" int rx_fg_row_update_clip(const float* d_compact, int32_t dim, const uint32_t* d_srow, int64_t n, int mode,
                                     float* d_w, float* d_state, const float* d_compact2, float* d_w2, float* d_state2,
                                     int64_t n_rows, float lr, float eps, const float* d_parts, int64_t n_parts,
                                     float max_norm, float* d_scale_out, float* d_norm_out, int64_t ld_w, int64_t ld_w2,
                                     int device, void* stream) {
  RX_REQUIRE(d_parts && n_parts >= 1 && d_scale_out && max_norm > 0.f, "rx_fg_row_update_clip: partial sums, their count, a positive norm bound and the scale output are required");
  RX_REQUIRE(n >= 1, "rx_fg_row_update_clip: an empty list leaves the clip factor unwritten; use rx_clip_scale_parts");
  return fg_row_update_run(d_compact, dim, d_srow, n, mode, d_w, d_state, d_compact2, d_w2, d_state2, n_rows, lr, eps, nullptr,
                           d_parts, n_parts, max_norm, d_scale_out, d_norm_out, ld_w, ld_w2, device, stream);
}

static int fg_row_update_run(const float* d_compact, int32_t dim, const uint32_t* d_srow, int64_t n, int mode,
                             float* d_w, float* d_state, const float* d_compact2, float* d_w2, float* d_state2,
                             int64_t n_rows, float lr, float eps, const float* d_scale, const float* d_parts,
                             int64_t n_parts, float max_norm, float* d_scale_out, float* d_norm_out, int64_t ld_w, int64_t ld_w2,
                             int device, void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(n >= 0 && fg_group_of(dim), "rx_fg_row_update: bad shape (dim 16, 32, 64 or 128)");
  RX_REQUIRE(mode >= RX_SEG_SGD && mode <= RX_SEG_SGD_SPEC_REDO, "rx_fg_row_update: unknown mode %d", mode);
  const bool force = mode == RX_SEG_SGD_SPEC_REDO;
  if (force) mode = RX_SEG_SGD_SPEC;
  const bool spec_line = mode == RX_SEG_SGD_SPEC && d_w2 && d_w2 == d_w + dim && (ld_w ? ld_w : dim) == 2 * dim && (ld_w2 ? ld_w2 : 1) == 2 * dim;
  RX_REQUIRE(spec_line ? d_compact2 == nullptr : (d_compact2 == nullptr) == (d_w2 == nullptr),
             "rx_fg_row_update: companion gradient and table come as a pair (the speculative step in the line layout keeps the pair in the line)");
  RX_REQUIRE(mode != RX_SEG_SGD_SPEC || d_scale || d_parts, "rx_fg_row_update: the speculative step needs the clip factor or the partial sums");
  if (n == 0) return 0;
  RX_REQUIRE(d_compact && d_srow && d_w && rx_aligned16(d_compact) && rx_aligned16(d_w) && rx_aligned16(d_srow), "rx_fg_row_update: null or unaligned argument");
  RX_REQUIRE(mode == RX_SEG_SGD || mode == RX_SEG_SGD_SPEC || (d_state && (!d_w2 || d_state2)), "rx_fg_row_update: Adagrad needs its state buffers");
  RX_REQUIRE(n_rows >= 1, "rx_fg_row_update: n_rows must be positive");
  FgRed a{};
  a.srow = d_srow; a.n = n; a.dim = dim; a.target = d_w; a.state = d_state; a.target2 = d_w2; a.state2 = d_state2;
  a.n_rows = n_rows; a.lr = lr; a.eps = eps; a.scale = d_scale;
  a.ldw = ld_w ? ld_w : dim; a.ldw2 = ld_w2 ? ld_w2 : 1;
  RX_REQUIRE(a.ldw >= dim && a.ldw % 4 == 0 && a.ldw2 >= 1, "rx_fg_row_update: table row stride must be a multiple of 4 floats and >= dim");
  a.line = (mode == RX_SEG_SGD || mode == RX_SEG_SGD_SPEC) && d_w2 && d_w2 == d_w + dim && a.ldw == 2 * dim && a.ldw2 == a.ldw;
  a.force = force ? 1 : 0;
  a.clip_parts = d_parts; a.clip_n = n_parts; a.clip_max = max_norm; a.clip_scale_out = d_scale_out; a.clip_norm_out = d_norm_out;
  hipStream_t st = (hipStream_t)stream;
  switch (fg_group_of(dim)) {
    case 4: FG_UPDATE_GROUP(4); break;
    case 8: FG_UPDATE_GROUP(8); break;
    case 16: FG_UPDATE_GROUP(16); break;
    default: FG_UPDATE_GROUP(32); break;
  }
  RX_LAUNCH_CHECK();
  return 0;
}
