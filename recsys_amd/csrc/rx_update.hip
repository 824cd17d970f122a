// Segmented gradient reduction fused with the sparse row update, plus the dense optimizers
// and norm reductions of the train step (gfx950).
//
// After rx_sort_dedup every distinct row owns a contiguous run of occurrence positions.  One
// lane group (dim/4 lanes, 16-byte accesses) walks the run in position order and accumulates
// in registers — a fixed order, so results are bitwise reproducible and no float atomic is
// issued — then applies the optimizer to the row it already holds.
#include "rx_common.h"

template <int VEC, int CH>
struct Acc {
  RxPack<VEC> p[CH];
};

#define ACC_FOREACH(k, i) \
  _Pragma("unroll") for (int k = 0; k < CH; ++k) _Pragma("unroll") for (int i = 0; i < VEC; ++i)

struct SegArgs {
  const float* occ;
  int64_t occ_ld;
  int64_t n;
  int dim;
  const int32_t* sorted_pos;
  const int32_t* sorted_slot;
  const int64_t* uniq_rows;
  const int32_t* seg_off;
  const int32_t* n_uniq;
  int mode;
  float* target;
  float* state;
  int64_t n_rows;
  float lr, eps;
  const float* scale;
  float* sq_part;
  int group;
  // companion scalar table addressed by the same rows (the D=1 bias table of FM / Wide&Deep)
  const float* occ2;
  float* target2;
  float* state2;
  // cross-piece workspace
  float* partials;    // [pieces][2][dim]
  float* partials2;   // [pieces][2]
  int32_t* tail_row;  // [pieces] slot of the run that leaves the piece to the right, or -1
};

// Optimizer arithmetic for one scalar (companion table entry).  MODE is a template parameter:
// the walk below reaches its flush code from every position of an unrolled batch, and a
// run-time switch over all optimizers at each site overflowed the instruction cache.
template <int MODE>
__device__ __forceinline__ void apply_scalar(const SegArgs& a, float g, int64_t row, int64_t u, float scale) {
  if constexpr (MODE == RX_SEG_DENSE) {
    a.target2[row] = g;
  } else if constexpr (MODE == RX_SEG_COMPACT) {
    a.target2[u] = g;
  } else if constexpr (MODE == RX_SEG_SGD) {
    a.target2[row] = a.target2[row] - a.lr * (g * scale);
  } else {  // Adagrad; row-wise Adagrad of a width-1 row is the same rule
    g *= scale;
    const float s = a.state2[row] + g * g;
    a.state2[row] = s;
    a.target2[row] = a.target2[row] - a.lr * (g / (sqrtf(s) + a.eps));
  }
}

template <int VEC, int CH, int MODE>
__device__ __forceinline__ void apply_row(const SegArgs& a, Acc<VEC, CH>& g, int64_t row, int64_t u, int lg,
                                          float scale) {
  const int dim = a.dim, group = a.group;
  if constexpr (MODE == RX_SEG_DENSE || MODE == RX_SEG_COMPACT) {
    float* dst = a.target + (MODE == RX_SEG_DENSE ? row : u) * (int64_t)dim;
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int c = (lg + k * group) * VEC;
      if (c < dim) rx_store<VEC>(dst + c, g.p[k]);
    }
    return;
  }
  ACC_FOREACH(k, i) g.p[k].v[i] *= scale;
  float* w = a.target + row * (int64_t)dim;
  if constexpr (MODE == RX_SEG_SGD) {
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int c = (lg + k * group) * VEC;
      if (c < dim) {
        RxPack<VEC> x = rx_load<VEC>(w + c);
#pragma unroll
        for (int i = 0; i < VEC; ++i) x.v[i] = x.v[i] - a.lr * g.p[k].v[i];
        rx_store<VEC>(w + c, x);
      }
    }
  } else if constexpr (MODE == RX_SEG_ADAGRAD) {
    float* st = a.state + row * (int64_t)dim;
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int c = (lg + k * group) * VEC;
      if (c < dim) {
        RxPack<VEC> x = rx_load<VEC>(w + c);
        RxPack<VEC> s = rx_load<VEC>(st + c);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          const float gi = g.p[k].v[i];
          s.v[i] += gi * gi;
          x.v[i] -= a.lr * (gi / (sqrtf(s.v[i]) + a.eps));
        }
        rx_store<VEC>(st + c, s);
        rx_store<VEC>(w + c, x);
      }
    }
  } else if constexpr (MODE == RX_SEG_ROWWISE_ADAGRAD) {
    float sq = 0.f;
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int c = (lg + k * group) * VEC;
      if (c < dim) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) sq += g.p[k].v[i] * g.p[k].v[i];
      }
    }
    sq = rx_group_sum(sq, group);
    const float s_new = a.state[row] + sq / (float)dim;
    const float step = a.lr / (sqrtf(s_new) + a.eps);
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int c = (lg + k * group) * VEC;
      if (c < dim) {
        RxPack<VEC> x = rx_load<VEC>(w + c);
#pragma unroll
        for (int i = 0; i < VEC; ++i) x.v[i] -= step * g.p[k].v[i];
        rx_store<VEC>(w + c, x);
      }
    }
    if (lg == 0) a.state[row] = s_new;
  }
}

// Runs are cut at fixed position boundaries (SEG_CHUNK sorted positions per lane group), so a
// row hit thousands of times (Zipf heads, tiny fields) costs no more wall time than a cold one:
//   pass 1  one lane group per piece.  The block first stages its (slot, position) tile in LDS
//           with one coalesced sweep; each group then walks its 32 positions in order, loading
//           occurrence rows a batch at a time so several HBM reads are in flight.  A run that
//           lies entirely inside the piece is finished and applied on the spot; a run that
//           crosses a piece boundary leaves a partial sum in partials[piece][0 = entered from the
//           left | 1 = leaves to the right], and the piece records the run that leaves it.
//   pass 2  one wavefront per piece whose tail run continues: its 64/group lane slots add the
//           run's partials (stride = slots, piece order inside a slot), a fixed xor tree joins
//           the slots, and slot 0 applies the row.
// The order of additions depends only on the sort, so results are bitwise reproducible.
#define SEG_CHUNK 32

template <int VEC, int CH>
__device__ __forceinline__ float acc_sq(const Acc<VEC, CH>& g, int lg, int group, int dim) {
  float sq = 0.f;
#pragma unroll
  for (int k = 0; k < CH; ++k) {
    const int c = (lg + k * group) * VEC;
    if (c < dim) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) sq += g.p[k].v[i] * g.p[k].v[i];
    }
  }
  return sq;
}

__device__ __forceinline__ void block_sq_store(float sq, float* sq_part, float* sq_lds) {
  sq = rx_wave_sum(sq);
  if ((threadIdx.x & 63) == 0) sq_lds[threadIdx.x >> 6] = sq;
  __syncthreads();
  if (threadIdx.x == 0) sq_part[blockIdx.x] = (sq_lds[0] + sq_lds[1]) + (sq_lds[2] + sq_lds[3]);
}

template <int VEC, int CH, int MODE>
__global__ void __launch_bounds__(RX_BLOCK)
segreduce_chunk_kernel(SegArgs a) {
  // LDS tile of the block's sorted positions, staged by one coalesced sweep:
  //   meta[1 + j] = (slot, position) of sorted position base + j; slot -1 past the end
  //   meta[0], meta[1 + T] = the neighbours just outside the tile (only their slot is used)
  //   rowid[j] = table row of that slot
  // so the walk below never waits on a dependent global load except the rows themselves.
  extern __shared__ int2 meta[];
  __shared__ float sq_lds[RX_BLOCK / 64];
  constexpr int BATCH = 8 / CH;
  const int group = a.group, dim = a.dim;
  const int gpb = RX_BLOCK / group;
  const int T = gpb * SEG_CHUNK;
  uint32_t* rowid = reinterpret_cast<uint32_t*>(meta + T + 2);
  const int64_t piece0 = (int64_t)blockIdx.x * gpb;
  {
    const int64_t base = piece0 * SEG_CHUNK;
    for (int j = threadIdx.x; j < T + 2; j += RX_BLOCK) {
      const int64_t p = base + j - 1;
      int2 m = make_int2(-1, 0);
      if (p >= 0 && p < a.n) m = make_int2(a.sorted_slot[p], a.sorted_pos[p]);
      meta[j] = m;
      if (j >= 1 && j <= T) rowid[j - 1] = m.x >= 0 ? (uint32_t)a.uniq_rows[m.x] : 0u;
    }
  }
  __syncthreads();
  const int my = threadIdx.x / group;
  const int lg = threadIdx.x % group;
  const int64_t piece = piece0 + my;
  const int64_t lo = piece * SEG_CHUNK;
  float sq = 0.f;
  if (lo < a.n) {
    const int len = (int)((lo + SEG_CHUNK < a.n) ? SEG_CHUNK : a.n - lo);
    const float scale = a.scale ? a.scale[0] : 1.f;
    const int2* mine = meta + 1 + my * SEG_CHUNK;  // mine[-1] and mine[len] are the neighbours
    const uint32_t* myrow = rowid + my * SEG_CHUNK;
    const int left = mine[-1].x, right = mine[len].x;
    Acc<VEC, CH> g;
    ACC_FOREACH(k, i) g.p[k].v[i] = 0.f;
    float g2 = 0.f;
    int cur = mine[0].x;
    int run_start = 0;
    int tail = -1;
    // flush the run of slot u that covers piece positions [run_start, end)
    auto flush = [&](int u, int end) {
      const bool entered = (run_start == 0) && (left == u);
      const bool leaves = (end == len) && (right == u);
      if (!entered && !leaves) {
        const int64_t row = (int64_t)myrow[run_start];
        if (a.sq_part) {
          sq += acc_sq<VEC, CH>(g, lg, group, dim);
          if (a.occ2 && lg == 0) sq += g2 * g2;
        }
        if (row < a.n_rows) {
          if (a.occ2 && lg == 0) apply_scalar<MODE>(a, g2, row, u, scale);
          apply_row<VEC, CH, MODE>(a, g, row, u, lg, scale);
        }
      } else {
        const int side = entered ? 0 : 1;
        float* dst = a.partials + (piece * 2 + side) * (int64_t)dim;
#pragma unroll
        for (int k = 0; k < CH; ++k) {
          const int c = (lg + k * group) * VEC;
          if (c < dim) rx_store<VEC>(dst + c, g.p[k]);
        }
        if (a.occ2 && lg == 0) a.partials2[piece * 2 + side] = g2;
        if (side == 1) tail = u;
      }
      ACC_FOREACH(k, i) g.p[k].v[i] = 0.f;
      g2 = 0.f;
      run_start = end;
    };
    // Loads are issued unconditionally (out-of-range batch entries re-read the piece's last
    // position, out-of-range columns re-read column 0) and masked afterwards: a load under a
    // run-time branch would make the compiler wait for each one before issuing the next.
    bool col_ok[CH];
    int col[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int c = (lg + k * group) * VEC;
      col_ok[k] = c < dim;
      col[k] = col_ok[k] ? c : 0;
    }
    const bool has2 = a.occ2 != nullptr;
#pragma unroll 1
    for (int b0 = 0; b0 < SEG_CHUNK; b0 += BATCH) {
      if (b0 >= len) break;
      int sl[BATCH];
      Acc<VEC, CH> rows[BATCH];
      float s2[BATCH];
#pragma unroll
      for (int j = 0; j < BATCH; ++j) {
        const int jj = (b0 + j < len) ? b0 + j : len - 1;
        const int2 m = mine[jj];
        sl[j] = m.x;
        const float* src = a.occ + (int64_t)m.y * a.occ_ld;
#pragma unroll
        for (int k = 0; k < CH; ++k) rows[j].p[k] = rx_load<VEC>(src + col[k]);
        s2[j] = has2 ? a.occ2[m.y] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < BATCH; ++j) {
        if (b0 + j < len) {
          if (sl[j] != cur) {
            flush(cur, b0 + j);
            cur = sl[j];
          }
#pragma unroll
          for (int k = 0; k < CH; ++k) {
            if (col_ok[k]) {
#pragma unroll
              for (int i = 0; i < VEC; ++i) g.p[k].v[i] += rows[j].p[k].v[i];
            }
          }
          g2 += s2[j];
        }
      }
    }
    flush(cur, len);
    if (lg == 0) a.tail_row[piece] = tail;
  }
  if (a.sq_part) block_sq_store(sq, a.sq_part, sq_lds);
}

template <int VEC, int CH, int MODE>
__global__ void __launch_bounds__(RX_BLOCK)
segreduce_combine_kernel(SegArgs a, int64_t n_pieces) {
  __shared__ float sq_lds[RX_BLOCK / 64];
  const int group = a.group, dim = a.dim;
  const int lane = threadIdx.x & 63;
  const int64_t piece = (int64_t)blockIdx.x * (RX_BLOCK / 64) + (threadIdx.x >> 6);
  float sq = 0.f;
  const int u = piece < n_pieces ? a.tail_row[piece] : -1;  // wave-uniform
  if (u >= 0) {
    const int64_t run_hi = a.seg_off[u + 1];
    const int64_t last = (run_hi - 1) / SEG_CHUNK;
    const int n_parts = (int)(last - piece) + 1;
    const int slots = 64 / group;
    const int s = lane / group, lg = lane % group;
    Acc<VEC, CH> g;
    ACC_FOREACH(k, i) g.p[k].v[i] = 0.f;
    float g2 = 0.f;
    for (int q = s; q < n_parts; q += slots) {
      const int64_t e = (piece + q) * 2 + (q == 0 ? 1 : 0);
      const float* src = a.partials + e * (int64_t)dim;
#pragma unroll
      for (int k = 0; k < CH; ++k) {
        const int c = (lg + k * group) * VEC;
        if (c < dim) {
          RxPack<VEC> x = rx_load<VEC>(src + c);
#pragma unroll
          for (int i = 0; i < VEC; ++i) g.p[k].v[i] += x.v[i];
        }
      }
      if (a.occ2 && lg == 0) g2 += a.partials2[e];
    }
    for (int m = group; m < 64; m <<= 1) {
      ACC_FOREACH(k, i) g.p[k].v[i] += __shfl_xor(g.p[k].v[i], m, 64);
      g2 += __shfl_xor(g2, m, 64);
    }
    if (s == 0) {
      const int64_t row = a.uniq_rows[u];
      if (a.sq_part) {
        sq = acc_sq<VEC, CH>(g, lg, group, dim);
        if (a.occ2 && lg == 0) sq += g2 * g2;
      }
      if (row >= 0 && row < a.n_rows) {
        const float scale = a.scale ? a.scale[0] : 1.f;
        if (a.occ2 && lg == 0) apply_scalar<MODE>(a, g2, row, u, scale);
        apply_row<VEC, CH, MODE>(a, g, row, u, lg, scale);
      }
    }
  }
  if (a.sq_part) block_sq_store(sq, a.sq_part, sq_lds);
}

// 16-byte accesses need 16-byte aligned rows; widths that are not a multiple of 4 use
// scalar lanes.  Launch grids are pure functions of (n, dim).
static int seg_shape(int dim, int64_t occ_ld, const void* occ, const void* target, const void* state,
                     RxRowShape* sh, const char* who) {
  *sh = rx_row_shape(dim);
  if (sh->vec == 4) {
    RX_REQUIRE((occ_ld % 4 == 0) && rx_aligned16(occ) && rx_aligned16(target) && (!state || rx_aligned16(state)),
               "%s: dim %d is a multiple of 4, so buffers must be 16-byte aligned and the row stride a multiple of 4", who, dim);
  }
  if (sh->chunks == 3) sh->chunks = 4;
  if (sh->group < 2) sh->group = 2;  // keeps the LDS index tile of pass 1 at <= 32 KiB
  RX_REQUIRE(sh->chunks <= 4, "%s: dim %d unsupported (above 256 it must be a multiple of 4)", who, dim);
  return 0;
}

static int seg_group(int dim) {
  const int g = rx_row_shape(dim).group;
  return g < 2 ? 2 : g;
}

static int64_t seg_pieces(int64_t n) { return rx_cdiv(n, SEG_CHUNK); }
static int64_t seg_grid1(int64_t n, int dim) { return rx_cdiv(seg_pieces(n) * seg_group(dim), RX_BLOCK); }
static int64_t seg_grid2(int64_t n) { return rx_cdiv(seg_pieces(n), RX_BLOCK / 64); }

extern "C" int64_t rx_segreduce_grid(int64_t n, int32_t dim) {
  if (n <= 0 || dim < 1 || dim > RX_MAX_DIM) return 0;
  return seg_grid1(n, dim) + seg_grid2(n);
}

extern "C" size_t rx_segreduce_workspace_bytes(int64_t n, int32_t dim) {
  if (n <= 0 || dim < 1) return 256;
  const size_t pieces = (size_t)seg_pieces(n);
  return rx_align_up(pieces * 2 * dim * sizeof(float), 256) + rx_align_up(pieces * 2 * sizeof(float), 256) +
         rx_align_up(pieces * sizeof(int32_t), 256);
}

#define SEG_DISPATCH_M(M, sh, KERNEL, grid, lds, st, ...)                                          \
  do {                                                                                             \
    if ((sh).vec == 4 && (sh).chunks == 1) hipLaunchKernelGGL((KERNEL<4, 1, M>), grid, dim3(RX_BLOCK), lds, st, __VA_ARGS__); \
    else if ((sh).vec == 4 && (sh).chunks == 2) hipLaunchKernelGGL((KERNEL<4, 2, M>), grid, dim3(RX_BLOCK), lds, st, __VA_ARGS__); \
    else if ((sh).vec == 4) hipLaunchKernelGGL((KERNEL<4, 4, M>), grid, dim3(RX_BLOCK), lds, st, __VA_ARGS__); \
    else if ((sh).chunks == 1) hipLaunchKernelGGL((KERNEL<1, 1, M>), grid, dim3(RX_BLOCK), lds, st, __VA_ARGS__); \
    else if ((sh).chunks == 2) hipLaunchKernelGGL((KERNEL<1, 2, M>), grid, dim3(RX_BLOCK), lds, st, __VA_ARGS__); \
    else hipLaunchKernelGGL((KERNEL<1, 4, M>), grid, dim3(RX_BLOCK), lds, st, __VA_ARGS__);       \
  } while (0)

#define SEG_DISPATCH(mode, sh, KERNEL, grid, lds, st, ...)                                         \
  do {                                                                                             \
    switch (mode) {                                                                                \
      case RX_SEG_DENSE: SEG_DISPATCH_M(RX_SEG_DENSE, sh, KERNEL, grid, lds, st, __VA_ARGS__); break; \
      case RX_SEG_COMPACT: SEG_DISPATCH_M(RX_SEG_COMPACT, sh, KERNEL, grid, lds, st, __VA_ARGS__); break; \
      case RX_SEG_SGD: SEG_DISPATCH_M(RX_SEG_SGD, sh, KERNEL, grid, lds, st, __VA_ARGS__); break;  \
      case RX_SEG_ADAGRAD: SEG_DISPATCH_M(RX_SEG_ADAGRAD, sh, KERNEL, grid, lds, st, __VA_ARGS__); break; \
      default: SEG_DISPATCH_M(RX_SEG_ROWWISE_ADAGRAD, sh, KERNEL, grid, lds, st, __VA_ARGS__); break; \
    }                                                                                              \
  } while (0)

static int mode_check(int mode, const float* state, const char* who, bool allow_grad_modes) {
  RX_REQUIRE(mode >= (allow_grad_modes ? RX_SEG_DENSE : RX_SEG_SGD) && mode <= RX_SEG_ROWWISE_ADAGRAD, "%s: unknown mode %d", who, mode);
  RX_REQUIRE(!(mode == RX_SEG_ADAGRAD || mode == RX_SEG_ROWWISE_ADAGRAD) || state, "%s: Adagrad needs a state buffer", who);
  return 0;
}

extern "C" int rx_segreduce(const float* d_occ, int64_t occ_ld, int64_t n, int32_t dim,
                            const int32_t* d_sorted_pos, const int32_t* d_sorted_slot,
                            const int64_t* d_uniq_rows, const int32_t* d_seg_off,
                            const int32_t* d_n_uniq, int mode, float* d_target, float* d_state,
                            const float* d_occ2, float* d_target2, float* d_state2, int64_t n_rows,
                            float lr, float eps, const float* d_scale, float* d_sq_part, void* ws,
                            size_t ws_bytes, int device, void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(n >= 0 && dim >= 1 && dim <= RX_MAX_DIM && occ_ld >= dim, "rx_segreduce: bad shape n=%lld dim=%d ld=%lld", (long long)n, dim, (long long)occ_ld);
  if (mode_check(mode, d_state, "rx_segreduce", true)) return 1;
  RX_REQUIRE((d_occ2 == nullptr) == (d_target2 == nullptr), "rx_segreduce: companion gradient and target come as a pair");
  RX_REQUIRE(!d_occ2 || !(mode == RX_SEG_ADAGRAD || mode == RX_SEG_ROWWISE_ADAGRAD) || d_state2, "rx_segreduce: companion Adagrad state missing");
  if (n == 0) return 0;
  RX_REQUIRE(d_occ && d_sorted_pos && d_sorted_slot && d_uniq_rows && d_seg_off && d_n_uniq && d_target && ws, "rx_segreduce: null argument");
  RX_REQUIRE(n_rows >= 1, "rx_segreduce: n_rows must be positive");
  RX_REQUIRE(ws_bytes >= rx_segreduce_workspace_bytes(n, dim) && rx_aligned16(ws), "rx_segreduce: workspace too small or unaligned");
  RxRowShape sh;
  if (seg_shape(dim, occ_ld, d_occ, d_target, mode == RX_SEG_ADAGRAD ? d_state : nullptr, &sh, "rx_segreduce")) return 1;
  const size_t pieces = (size_t)seg_pieces(n);
  char* w = (char*)ws;
  float* partials = (float*)w; w += rx_align_up(pieces * 2 * dim * sizeof(float), 256);
  float* partials2 = (float*)w; w += rx_align_up(pieces * 2 * sizeof(float), 256);
  int32_t* tail_row = (int32_t*)w;
  SegArgs a{d_occ, occ_ld, n, dim, d_sorted_pos, d_sorted_slot, d_uniq_rows, d_seg_off, d_n_uniq, mode, d_target,
            d_state, n_rows, lr, eps, d_scale, d_sq_part, sh.group, d_occ2, d_target2, d_state2, partials,
            partials2, tail_row};
  const unsigned g1 = (unsigned)seg_grid1(n, dim), g2 = (unsigned)seg_grid2(n);
  const size_t tile = (size_t)(RX_BLOCK / sh.group) * SEG_CHUNK;
  const size_t lds = (tile + 2) * sizeof(int2) + tile * sizeof(uint32_t);
  hipStream_t st = (hipStream_t)stream;
  SEG_DISPATCH(mode, sh, segreduce_chunk_kernel, dim3(g1), lds, st, a);
  RX_LAUNCH_CHECK();
  if (d_sq_part) a.sq_part = d_sq_part + g1;
  SEG_DISPATCH(mode, sh, segreduce_combine_kernel, dim3(g2), 0, st, a, (int64_t)pieces);
  RX_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- update from a compact gradient
template <int VEC, int CH, int MODE>
__global__ void __launch_bounds__(RX_BLOCK)
row_update_kernel(SegArgs a) {
  const int group = a.group, dim = a.dim;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t u = gid / group;
  const int lg = (int)(gid % group);
  if (u >= a.n_uniq[0]) return;
  const int64_t row = a.uniq_rows[u];
  if (row < 0 || row >= a.n_rows) return;
  Acc<VEC, CH> g;
#pragma unroll
  for (int k = 0; k < CH; ++k) {
    const int c = (lg + k * group) * VEC;
    g.p[k] = (c < dim) ? rx_load<VEC>(a.occ + u * (int64_t)dim + c) : rx_zero<VEC>();
  }
  apply_row<VEC, CH, MODE>(a, g, row, u, lg, a.scale ? a.scale[0] : 1.f);
}

extern "C" int rx_row_update(const float* d_compact, int32_t dim, const int64_t* d_uniq_rows,
                             const int32_t* d_n_uniq, int64_t n_max, int mode, float* d_w,
                             float* d_state, int64_t n_rows, float lr, float eps,
                             const float* d_scale, int device, void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(n_max >= 0 && dim >= 1 && dim <= RX_MAX_DIM, "rx_row_update: bad shape");
  if (mode_check(mode, d_state, "rx_row_update", false)) return 1;
  if (n_max == 0) return 0;
  RX_REQUIRE(d_compact && d_uniq_rows && d_n_uniq && d_w && n_rows >= 1, "rx_row_update: null argument");
  RxRowShape sh;
  if (seg_shape(dim, dim, d_compact, d_w, mode == RX_SEG_ADAGRAD ? d_state : nullptr, &sh, "rx_row_update")) return 1;
  SegArgs a{d_compact, dim, n_max, dim, nullptr, nullptr, d_uniq_rows, nullptr, d_n_uniq, mode, d_w, d_state,
            n_rows, lr, eps, d_scale, nullptr, sh.group, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  const unsigned grid = (unsigned)rx_cdiv(n_max * sh.group, RX_BLOCK);
  SEG_DISPATCH(mode, sh, row_update_kernel, dim3(grid), 0, (hipStream_t)stream, a);
  RX_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- dense optimizers
static inline unsigned dense_grid(int64_t n) {
  int64_t g = rx_cdiv(n, RX_BLOCK);
  return (unsigned)(g < 2048 ? g : 2048);
}

__global__ void __launch_bounds__(RX_BLOCK)
dense_adam_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m,
                  float* __restrict__ v, int64_t n, float beta1, float beta2, float eps,
                  float step_size, float inv_sqrt_bc2, const float* __restrict__ scale) {
  const float sc = scale ? scale[0] : 1.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[i] * sc;
    const float mi = m[i] + (gi - m[i]) * (1.f - beta1);
    const float vi = v[i] * beta2 + (1.f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
    w[i] = w[i] - step_size * (mi / denom);
  }
}

extern "C" int rx_dense_adam(float* d_w, const float* d_g, float* d_m, float* d_v, int64_t n, float lr,
                             float beta1, float beta2, float eps, int32_t step, const float* d_scale,
                             int device, void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(n >= 0 && step >= 1, "rx_dense_adam: bad n or step");
  if (n == 0) return 0;
  RX_REQUIRE(d_w && d_g && d_m && d_v, "rx_dense_adam: null argument");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  hipLaunchKernelGGL(dense_adam_kernel, dim3(dense_grid(n)), dim3(RX_BLOCK), 0, (hipStream_t)stream, d_w, d_g, d_m, d_v, n, beta1, beta2, eps, step_size, inv_sqrt_bc2, d_scale);
  RX_LAUNCH_CHECK();
  return 0;
}

__global__ void __launch_bounds__(RX_BLOCK)
dense_sgd_kernel(float* __restrict__ w, const float* __restrict__ g, int64_t n, float lr, const float* __restrict__ scale) {
  const float sc = scale ? scale[0] : 1.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    w[i] = w[i] - lr * (g[i] * sc);
}

extern "C" int rx_dense_sgd(float* d_w, const float* d_g, int64_t n, float lr, const float* d_scale,
                            int device, void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(n >= 0, "rx_dense_sgd: bad n");
  if (n == 0) return 0;
  RX_REQUIRE(d_w && d_g, "rx_dense_sgd: null argument");
  hipLaunchKernelGGL(dense_sgd_kernel, dim3(dense_grid(n)), dim3(RX_BLOCK), 0, (hipStream_t)stream, d_w, d_g, n, lr, d_scale);
  RX_LAUNCH_CHECK();
  return 0;
}

__global__ void __launch_bounds__(RX_BLOCK)
dense_adagrad_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ s, int64_t n,
                     float lr, float eps, const float* __restrict__ scale) {
  const float sc = scale ? scale[0] : 1.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[i] * sc;
    const float si = s[i] + gi * gi;
    s[i] = si;
    w[i] = w[i] - lr * (gi / (sqrtf(si) + eps));
  }
}

extern "C" int rx_dense_adagrad(float* d_w, const float* d_g, float* d_state, int64_t n, float lr,
                                float eps, const float* d_scale, int device, void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(n >= 0, "rx_dense_adagrad: bad n");
  if (n == 0) return 0;
  RX_REQUIRE(d_w && d_g && d_state, "rx_dense_adagrad: null argument");
  hipLaunchKernelGGL(dense_adagrad_kernel, dim3(dense_grid(n)), dim3(RX_BLOCK), 0, (hipStream_t)stream, d_w, d_g, d_state, n, lr, eps, d_scale);
  RX_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- reductions for the global clip
#define RX_RED_MAX_BLOCKS 1024

template <bool SQUARE>
__global__ void __launch_bounds__(RX_BLOCK)
reduce_partial_kernel(const float* __restrict__ x, int64_t n, double* __restrict__ partial) {
  __shared__ double wsum[RX_BLOCK / 64];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double v = (double)x[i];
    acc += SQUARE ? v * v : v;
  }
  acc = rx_wave_sum_f64(acc);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

// mode 0: acc[0] += sum(partial)   mode 1: out_f32[0] = (float)sum(partial)
// one wavefront: lane l adds partial[l], partial[l+64], ... in order, then a fixed xor tree
__global__ void reduce_final_kernel(const double* __restrict__ partial, int n, double* __restrict__ acc,
                                    float* __restrict__ out_f32) {
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) s += partial[i];
  s = rx_wave_sum_f64(s);
  if (threadIdx.x != 0) return;
  if (acc) acc[0] += s;
  if (out_f32) out_f32[0] = (float)s;
}

extern "C" size_t rx_sumsq_workspace_bytes(int64_t n) {
  (void)n;
  return RX_RED_MAX_BLOCKS * sizeof(double);
}

extern "C" int rx_sumsq_accumulate(const float* d_x, int64_t n, double* d_acc, void* ws, size_t ws_bytes,
                                   int device, void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(n >= 0 && d_acc, "rx_sumsq_accumulate: bad arguments");
  if (n == 0) return 0;
  RX_REQUIRE(d_x && ws && ws_bytes >= rx_sumsq_workspace_bytes(n), "rx_sumsq_accumulate: null pointer or small workspace");
  const int grid = (int)std::min<int64_t>(rx_cdiv(n, RX_BLOCK), RX_RED_MAX_BLOCKS);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(reduce_partial_kernel<true>, dim3(grid), dim3(RX_BLOCK), 0, st, d_x, n, (double*)ws);
  RX_LAUNCH_CHECK();
  hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(64), 0, st, (const double*)ws, grid, d_acc, (float*)nullptr);
  RX_LAUNCH_CHECK();
  return 0;
}

extern "C" int rx_sum_accumulate(const float* d_x, int64_t n, double* d_acc, void* ws, size_t ws_bytes,
                                 int device, void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(n >= 0 && d_acc, "rx_sum_accumulate: bad arguments");
  if (n == 0) return 0;
  RX_REQUIRE(d_x && ws && ws_bytes >= rx_sumsq_workspace_bytes(n), "rx_sum_accumulate: null pointer or small workspace");
  const int grid = (int)std::min<int64_t>(rx_cdiv(n, RX_BLOCK), RX_RED_MAX_BLOCKS);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(reduce_partial_kernel<false>, dim3(grid), dim3(RX_BLOCK), 0, st, d_x, n, (double*)ws);
  RX_LAUNCH_CHECK();
  hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(64), 0, st, (const double*)ws, grid, d_acc, (float*)nullptr);
  RX_LAUNCH_CHECK();
  return 0;
}

extern "C" int rx_reduce_sum(const float* d_x, int64_t n, float* d_out, void* ws, size_t ws_bytes,
                             int device, void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(n >= 1 && d_x && d_out && ws && ws_bytes >= rx_sumsq_workspace_bytes(n), "rx_reduce_sum: bad arguments");
  const int grid = (int)std::min<int64_t>(rx_cdiv(n, RX_BLOCK), RX_RED_MAX_BLOCKS);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(reduce_partial_kernel<false>, dim3(grid), dim3(RX_BLOCK), 0, st, d_x, n, (double*)ws);
  RX_LAUNCH_CHECK();
  hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(64), 0, st, (const double*)ws, grid, (double*)nullptr, d_out);
  RX_LAUNCH_CHECK();
  return 0;
}

__global__ void clip_scale_kernel(const double* __restrict__ acc, float max_norm, float* __restrict__ scale,
                                  float* __restrict__ total_norm) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float tn = (float)sqrt(acc[0]);
  const float coef = max_norm / (tn + 1e-6f);
  scale[0] = coef < 1.f ? coef : 1.f;
  if (total_norm) total_norm[0] = tn;
}

extern "C" int rx_clip_scale(const double* d_acc, float max_norm, float* d_scale, float* d_total_norm,
                             int device, void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(d_acc && d_scale, "rx_clip_scale: null argument");
  hipLaunchKernelGGL(clip_scale_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d_acc, max_norm, d_scale, d_total_norm);
  RX_LAUNCH_CHECK();
  return 0;
}
