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
  // FM-fused gradient synthesis (N = 1 factorization machine, see rx_segreduce_fm):
  //   occurrence at list position p belongs to sample b(p) = p < fm_split ? p / fm_div : p - fm_split,
  //   its gradient row is fm_g[b] * (occ[b, :] - table[row, :]); nothing per-occurrence is stored
  const float* fm_g;      // [samples] dL/dlogit
  const float* fm_table;  // table to read the old row from when `target` is a gradient buffer
  int64_t fm_split;
  int fm_div;
  // optional tower-input gradient [samples, fm_deep_ld] laid out [item row, field 0 .. F-1 rows, ...]:
  // the occurrence of field f additionally receives fm_deep[b, (1 + f) * dim ...], the item's [b, 0 ...]
  const float* fm_deep;
  int64_t fm_deep_ld;
  // cross-piece workspace
  float* partials;    // [pieces][2][dim]
  float* partials2;   // [pieces][2]
  int32_t* tail_row;  // [pieces] slot of the run that leaves the piece to the right, or -1
  // RX_SEG_SGD_SPEC (the clipped SGD step taken by the reduce with a factor of 1): `state` holds the undo
  // records [n][2 * dim] = gradient | row before the step, `state2` [n][2] = companion gradient | entry before
  // the step, both indexed like a compact gradient.  In rx_row_update: redo even for a factor of 1 (`force`),
  // and the number of 256-thread blocks the list needs (the launch itself is a small grid that strides).
  int force;
  unsigned upd_blocks;
  // RX_SEG_SGD_SPEC_OLD: the records hold only what was there before the step -- `state` [n][dim] the row,
  // `state2` [n] the companion entry (half the bytes; a step whose clip bites is then redone from scratch: rows put
  // back, gradients formed again).  In rx_row_update the records are copied back (RX_SEG_RESTORE*).
  int undo_old_only;
  // *_IF_CLIPPED modes: the whole launch returns at once when this device scalar (the clip factor) is exactly 1
  const float* gate;
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
  } else if constexpr (MODE == RX_SEG_SGD_SPEC) {
    const float old = a.target2[row];
    if (a.undo_old_only) a.state2[u] = old;
    else reinterpret_cast<float2*>(a.state2)[u] = make_float2(g, old);
    a.target2[row] = old - a.lr * g;
  } else {  // Adagrad; row-wise Adagrad of a width-1 row is the same rule
    g *= scale;
    const float s = a.state2[row] + g * g;
    a.state2[row] = s;
    a.target2[row] = a.target2[row] - a.lr * (g / (sqrtf(s) + a.eps));
  }
}

// Writes the result for one row.  `x` is the row's current value and `st` its Adagrad state,
// loaded by the caller ahead of time (the walk prefetches them with the occurrence rows so that
// a flush never waits on a dependent HBM read); both are ignored by the gradient modes.
template <int VEC, int CH, int MODE>
__device__ __forceinline__ void apply_row(const SegArgs& a, Acc<VEC, CH>& g, Acc<VEC, CH>& x, Acc<VEC, CH>& st,
                                          int64_t row, int64_t u, int lg, float scale) {
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
  float* w = a.target + row * (int64_t)dim;
  if constexpr (MODE == RX_SEG_SGD_SPEC) {
    // the step with a clip factor of 1 (exact: the bits of the clipped step whenever the norm stays inside the
    // bound), and the record from which rx_row_update redoes the row when it does not
    if (a.undo_old_only) {
      float* ud = a.state + u * (int64_t)dim;
#pragma unroll
      for (int k = 0; k < CH; ++k) {
        const int c = (lg + k * group) * VEC;
        if (c < dim) rx_store<VEC>(ud + c, x.p[k]);
      }
    } else {
      float* ud = a.state + u * (int64_t)(2 * dim);
#pragma unroll
      for (int k = 0; k < CH; ++k) {
        const int c = (lg + k * group) * VEC;
        if (c < dim) {
          rx_store<VEC>(ud + c, g.p[k]);
          rx_store<VEC>(ud + dim + c, x.p[k]);
        }
      }
    }
    ACC_FOREACH(k, i) x.p[k].v[i] = x.p[k].v[i] - a.lr * g.p[k].v[i];
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int c = (lg + k * group) * VEC;
      if (c < dim) rx_store<VEC>(w + c, x.p[k]);
    }
    return;
  }
  ACC_FOREACH(k, i) g.p[k].v[i] *= scale;
  if constexpr (MODE == RX_SEG_SGD) {
    ACC_FOREACH(k, i) x.p[k].v[i] = x.p[k].v[i] - a.lr * g.p[k].v[i];
  } else if constexpr (MODE == RX_SEG_ADAGRAD) {
    ACC_FOREACH(k, i) {
      const float gi = g.p[k].v[i];
      st.p[k].v[i] += gi * gi;
      x.p[k].v[i] -= a.lr * (gi / (sqrtf(st.p[k].v[i]) + a.eps));
    }
    float* sp = a.state + row * (int64_t)dim;
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int c = (lg + k * group) * VEC;
      if (c < dim) rx_store<VEC>(sp + c, st.p[k]);
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
    ACC_FOREACH(k, i) x.p[k].v[i] -= step * g.p[k].v[i];
    if (lg == 0) a.state[row] = s_new;
  }
#pragma unroll
  for (int k = 0; k < CH; ++k) {
    const int c = (lg + k * group) * VEC;
    if (c < dim) rx_store<VEC>(w + c, x.p[k]);
  }
}

// Loads the row's current value (and Adagrad state) for apply_row; columns past dim read column 0.
template <int VEC, int CH, int MODE, bool FM>
__device__ __forceinline__ void load_old(const SegArgs& a, int64_t row, int lg, Acc<VEC, CH>& x, Acc<VEC, CH>& st) {
  constexpr bool OPT = MODE >= RX_SEG_SGD;
  if constexpr (FM || OPT) {
    const float* base = (OPT ? a.target : a.fm_table) + row * (int64_t)a.dim;
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int c = (lg + k * a.group) * VEC;
      x.p[k] = rx_load<VEC>(base + (c < a.dim ? c : 0));
    }
  }
  if constexpr (MODE == RX_SEG_ADAGRAD) {
    const float* base = a.state + row * (int64_t)a.dim;
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int c = (lg + k * a.group) * VEC;
      st.p[k] = rx_load<VEC>(base + (c < a.dim ? c : 0));
    }
  }
}

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

// Everything that happens once a row's summed gradient is known.  FM: the sum so far is
// sum_b g_b T_b; the row's own term (sum_b g_b) * e_row is subtracted here, where the old row is
// read anyway for the update.
template <int VEC, int CH, int MODE, bool FM>
__device__ __forceinline__ float finish_run(const SegArgs& a, Acc<VEC, CH>& g, float g2, Acc<VEC, CH>& x,
                                            Acc<VEC, CH>& st, int64_t row, int64_t u, int lg, float scale,
                                            bool has2) {
  const int dim = a.dim, group = a.group;
  if constexpr (FM) ACC_FOREACH(k, i) g.p[k].v[i] -= g2 * x.p[k].v[i];
  float sq = 0.f;
  if (a.sq_part) {
    sq = acc_sq<VEC, CH>(g, lg, group, dim);
    if (has2 && lg == 0) sq += g2 * g2;
  }
  if (has2 && lg == 0) apply_scalar<MODE>(a, g2, row, u, scale);
  apply_row<VEC, CH, MODE>(a, g, x, st, row, u, lg, scale);
  return sq;
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
#ifndef SEG_CHUNK
#define SEG_CHUNK 32
#endif

__device__ __forceinline__ void block_sq_store(float sq, float* sq_part, float* sq_lds, unsigned blk) {
  sq = rx_wave_sum(sq);
  if ((threadIdx.x & 63) == 0) sq_lds[threadIdx.x >> 6] = sq;
  __syncthreads();
  if (threadIdx.x == 0) sq_part[blk] = (sq_lds[0] + sq_lds[1]) + (sq_lds[2] + sq_lds[3]);
}

template <int VEC, int CH, int MODE, bool FM, bool DEEP>
__device__ __forceinline__ void segreduce_chunk_body(const SegArgs& a, const unsigned blk) {
  // LDS tile of the block's sorted positions, staged by one coalesced sweep.  Each piece owns
  // PS = SEG_CHUNK + 1 consecutive entries: (slot, position) of its positions (slot -1 past the end
  // of the list) and, in the extra entry, the slot of its right neighbour.  The odd stride puts
  // the pieces that one wavefront reads together on different LDS banks.  meta[0] is the entry
  // just left of the tile, and rowid[] holds each entry's table row with the same stride, so the
  // walk below never waits on a dependent global load except the rows themselves.
  extern __shared__ int2 meta[];
  __shared__ float sq_lds[RX_BLOCK / 64];
  constexpr int BATCH = 4 / CH;
  constexpr int PS = SEG_CHUNK + 1;
  const int group = a.group, dim = a.dim;
  const int gpb = RX_BLOCK / group;
  const int T = gpb * SEG_CHUNK;
  int2* tile = meta + 2;
  uint32_t* rowid = reinterpret_cast<uint32_t*>(tile + gpb * PS);
  uint32_t* deepoff = rowid + gpb * PS;   // DEEP only: element offset of the occurrence's slice of fm_deep
  const int64_t piece0 = (int64_t)blk * gpb;
  {
    // Eight entries per thread and round: all index loads of a round are issued before the
    // dependent row-id gathers, and those before the LDS stores, so a round costs two memory
    // latencies instead of two per entry.
    const int64_t base = piece0 * SEG_CHUNK;
    for (int j0 = 0; j0 < T + 2; j0 += 8 * RX_BLOCK) {
      int sl[8], ps[8];
      uint32_t rw[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int j = j0 + r * RX_BLOCK + threadIdx.x;
        int64_t p = base + j - 1;
        const bool ok = j < T + 2 && p >= 0 && p < a.n;
        p = ok ? p : 0;
        sl[r] = a.sorted_slot[p];
        ps[r] = a.sorted_pos[p];
        if (!ok) sl[r] = -1;
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) rw[r] = (uint32_t)a.uniq_rows[sl[r] >= 0 ? sl[r] : 0];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int j = j0 + r * RX_BLOCK + threadIdx.x;
        if (j < T + 2) {
          int pos = ps[r];
          uint32_t doff = 0;
          if (FM) {
            const bool ex = pos < a.fm_split;
            const int b = ex ? pos / a.fm_div : pos - (int)a.fm_split;  // sample of the occurrence
            if (DEEP) doff = (uint32_t)((int64_t)b * a.fm_deep_ld + (ex ? (1 + pos - b * a.fm_div) * a.dim : 0));
            pos = b;
          }
          const int2 e = make_int2(sl[r], sl[r] >= 0 ? pos : 0);
          const int t = j - 1;
          if (t < 0) {
            meta[0] = e;
          } else {
            const int pc = t / SEG_CHUNK, off = t % SEG_CHUNK;
            if (t < T) {
              tile[pc * PS + off] = e;
              rowid[pc * PS + off] = sl[r] >= 0 ? rw[r] : 0u;
              if (DEEP) deepoff[pc * PS + off] = sl[r] >= 0 ? doff : 0u;
            }
            if (off == 0 && t > 0) tile[(pc - 1) * PS + SEG_CHUNK] = e;
          }
        }
      }
    }
  }
  __syncthreads();
  const int my = threadIdx.x / group;
  const int lg = threadIdx.x % group;
  const int64_t piece = piece0 + my;
  const int64_t lo = piece * SEG_CHUNK;
  float sq = 0.f;
  if (lo < a.n) {
    // Positions past the end of the list carry slot -1 in the tile; they form a last "run" that is
    // never written, so the walk needs no length test per position.
    const int len = (int)((lo + SEG_CHUNK < a.n) ? SEG_CHUNK : a.n - lo);
    const float scale = a.scale ? a.scale[0] : 1.f;
    const int2* mine = tile + my * PS;  // mine[-2] (the previous piece's last entry) and mine[SEG_CHUNK] are the neighbours
    const uint32_t* myrow = rowid + my * PS;
    const uint32_t* mydeep = deepoff + my * PS;
    const int left = mine[-2].x, right = mine[SEG_CHUNK].x;
    Acc<VEC, CH> g;
    ACC_FOREACH(k, i) g.p[k].v[i] = 0.f;
    float g2 = 0.f;
    const bool has2 = a.target2 != nullptr;     // a companion table is updated
    const bool keep2 = FM || has2;              // the scalar sum is needed (FM: for the row's own term)
    const float* src2 = FM ? a.fm_g : a.occ2;
    // the companion's current value rides along with the rows when the update needs it
    constexpr bool OLD2 = MODE == RX_SEG_SGD;
    const float* old2_src = has2 ? a.target2 : a.occ;
    float* part0 = a.partials + piece * 2 * (int64_t)dim;   // [0] entered from the left, [1] leaves to the right
    float* part0_2 = a.partials2 + piece * 2;
    // Loads are issued unconditionally (out-of-range columns re-read column 0) and masked
    // afterwards: a load under a run-time branch would make the compiler wait for each one
    // before issuing the next.
    bool col_ok[CH];
    int col[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int c = (lg + k * group) * VEC;
      col_ok[k] = c < dim;
      col[k] = col_ok[k] ? c : 0;
    }
    // Two batches alternate: the loads of the next one are in flight while the current one is
    // walked, so a piece pays the memory latency once instead of once per batch.
    struct Batch {
      int sl[BATCH + 1];
      uint32_t rid[BATCH];
      Acc<VEC, CH> rows[BATCH], oldx[BATCH], oldst[BATCH], deep[DEEP ? BATCH : 1];
      float s2[BATCH], old2[BATCH];
    };
    auto load_batch = [&](int b0, Batch& t) {
#pragma unroll
      for (int j = 0; j < BATCH; ++j) {
        const int2 m = mine[b0 + j];
        t.sl[j] = m.x;
        const float* src = a.occ + (int64_t)m.y * a.occ_ld;
#pragma unroll
        for (int k = 0; k < CH; ++k) t.rows[j].p[k] = rx_load<VEC>(src + col[k]);
        t.s2[j] = keep2 ? src2[m.y] : 0.f;
        if constexpr (DEEP) {
          const float* dsrc = a.fm_deep + mydeep[b0 + j];
#pragma unroll
          for (int k = 0; k < CH; ++k) t.deep[j].p[k] = rx_load<VEC>(dsrc + col[k]);
        }
        // the row's current value (and state) ride along, so that the end of a run has them at hand.  Only
        // the last position of a run uses them: every other position reads row 0 instead (one line the
        // whole launch shares), which keeps the load unconditional without moving the row through the
        // caches once per occurrence.
        const uint32_t r = myrow[b0 + j];
        t.rid[j] = r;
        const bool ends = mine[b0 + j + 1].x != m.x;
        const int64_t r_safe = (ends && r < a.n_rows) ? (int64_t)r : 0;
        load_old<VEC, CH, MODE, FM>(a, r_safe, lg, t.oldx[j], t.oldst[j]);
        t.old2[j] = OLD2 ? old2_src[has2 ? r_safe : 0] : 0.f;
      }
      t.sl[BATCH] = mine[b0 + BATCH].x;
    };
    auto walk_batch = [&](int b0, Batch& t) {
      // Retire every load of the batch here.  vmcnt counts loads and stores in one queue, and the
      // run ends below store under run-time predicates; a load still pending behind such a store
      // would make each later use wait for the store's round trip as well.
#pragma unroll
      for (int j = 0; j < BATCH; ++j) {
        asm volatile("" ::"v"(t.s2[j]));
        if constexpr (OLD2) asm volatile("" ::"v"(t.old2[j]));
        ACC_FOREACH(k, i) {
          asm volatile("" ::"v"(t.rows[j].p[k].v[i]));
          if constexpr (DEEP) asm volatile("" ::"v"(t.deep[j].p[k].v[i]));
          if constexpr (FM || MODE >= RX_SEG_SGD) asm volatile("" ::"v"(t.oldx[j].p[k].v[i]));
          if constexpr (MODE == RX_SEG_ADAGRAD) asm volatile("" ::"v"(t.oldst[j].p[k].v[i]));
        }
      }
#pragma unroll
      for (int j = 0; j < BATCH; ++j) {
#pragma unroll
        for (int k = 0; k < CH; ++k) {
          if (col_ok[k]) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
              if constexpr (DEEP) g.p[k].v[i] += t.deep[j].p[k].v[i];
              if constexpr (FM) g.p[k].v[i] += t.s2[j] * t.rows[j].p[k].v[i];
              else g.p[k].v[i] += t.rows[j].p[k].v[i];
            }
          }
        }
        g2 += t.s2[j];
        if (t.sl[j + 1] != t.sl[j]) {
          // The run of slot u ends here.  It is complete unless it came in from the left
          // neighbour (a run that also leaves to the right never ends inside the piece).
          const int u = t.sl[j];
          const bool complete = u != left;
          const int64_t row = (int64_t)t.rid[j];
          const bool row_ok = row < a.n_rows;
          if constexpr (MODE == RX_SEG_DENSE || MODE == RX_SEG_COMPACT || MODE == RX_SEG_SGD) {
            // one store whose address and value are selected, instead of two code paths
            Acc<VEC, CH> G = g;
            if constexpr (FM) ACC_FOREACH(k, i) G.p[k].v[i] -= g2 * t.oldx[j].p[k].v[i];
            if (a.sq_part) {
              float q = acc_sq<VEC, CH>(G, lg, group, dim);
              if (has2 && lg == 0) q += g2 * g2;
              sq += (complete && row_ok) ? q : 0.f;
            }
            float v2 = g2;
            if constexpr (MODE == RX_SEG_SGD) {
              ACC_FOREACH(k, i) G.p[k].v[i] = t.oldx[j].p[k].v[i] - a.lr * (G.p[k].v[i] * scale);
              v2 = t.old2[j] - a.lr * (g2 * scale);
            }
            const int64_t idx = MODE == RX_SEG_COMPACT ? (int64_t)u : row;
            float* dst = complete ? a.target + idx * (int64_t)dim : part0;
            if (complete ? row_ok : u >= 0) {
#pragma unroll
              for (int k = 0; k < CH; ++k) {
                RxPack<VEC> val;
#pragma unroll
                for (int i = 0; i < VEC; ++i) val.v[i] = complete ? G.p[k].v[i] : g.p[k].v[i];
                if (col_ok[k]) rx_store<VEC>(dst + (lg + k * group) * VEC, val);
              }
            }
            if (keep2 && lg == 0 && (complete ? (has2 && row_ok) : u >= 0)) {
              float* dst2 = complete ? a.target2 + idx : part0_2;
              *dst2 = complete ? v2 : g2;
            }
          } else {
            if (complete) {
              if (row_ok) sq += finish_run<VEC, CH, MODE, FM>(a, g, g2, t.oldx[j], t.oldst[j], row, u, lg, scale, has2);
            } else if (u >= 0) {
#pragma unroll
              for (int k = 0; k < CH; ++k)
                if (col_ok[k]) rx_store<VEC>(part0 + (lg + k * group) * VEC, g.p[k]);
              if (keep2 && lg == 0) part0_2[0] = g2;
            }
          }
          ACC_FOREACH(k, i) g.p[k].v[i] = 0.f;
          g2 = 0.f;
        }
      }
    };
    Batch ba, bb;
    load_batch(0, ba);
#pragma unroll 1
    for (int b0 = 0; b0 < SEG_CHUNK; b0 += 2 * BATCH) {
      if (b0 >= len) break;
      load_batch(b0 + BATCH, bb);
      walk_batch(b0, ba);
      if (b0 + 2 * BATCH < SEG_CHUNK) load_batch(b0 + 2 * BATCH, ba);
      walk_batch(b0 + BATCH, bb);
    }
    // the run that leaves the piece to the right: side 0 if it also came in from the left
    const int last = mine[SEG_CHUNK - 1].x;
    int tail = -1;
    if (last >= 0 && last == right) {
      const int side = last == left ? 0 : 1;
#pragma unroll
      for (int k = 0; k < CH; ++k)
        if (col_ok[k]) rx_store<VEC>(part0 + side * dim + (lg + k * group) * VEC, g.p[k]);
      if (keep2 && lg == 0) part0_2[side] = g2;
      if (side == 1) tail = last;
    }
    if (lg == 0) a.tail_row[piece] = tail;
  }
  if (a.sq_part) block_sq_store(sq, a.sq_part, sq_lds, blk);
}

template <int VEC, int CH, int MODE, bool FM>
__global__ void __launch_bounds__(RX_BLOCK)
segreduce_chunk_kernel(SegArgs a) {
  if (a.gate && a.gate[0] == 1.f) return;
  segreduce_chunk_body<VEC, CH, MODE, FM, false>(a, blockIdx.x);
}

// Several tables in one launch (the user / item tables of the MF and NCF families: a few thousand
// occurrences each, where a launch per table costs more than the work).  Each table keeps its own
// blocks, pieces, workspace and partials, so results are those of one launch per table, bit for bit.
#define SEG_MULTI_MAX 8
struct SegMulti {
  SegArgs t[SEG_MULTI_MAX];
  int64_t pieces[SEG_MULTI_MAX];
  unsigned first[SEG_MULTI_MAX + 1];  // first block of each table; first[count] = grid size
  int count;
};

__device__ __forceinline__ int seg_multi_table(const SegMulti& m) {
  int t = 0;
  while (t + 1 < m.count && blockIdx.x >= m.first[t + 1]) ++t;
  return t;
}

template <int VEC, int CH, int MODE, bool FM>
__global__ void __launch_bounds__(RX_BLOCK)
segreduce_chunk_multi_kernel(SegMulti m) {
  const int t = seg_multi_table(m);
  segreduce_chunk_body<VEC, CH, MODE, false, false>(m.t[t], blockIdx.x - m.first[t]);
}

// FM synthesis plus the tower-input gradient slice of every occurrence (FM is implied)
template <int VEC, int CH, int MODE, bool FM>
__global__ void __launch_bounds__(RX_BLOCK)
segreduce_chunk_deep_kernel(SegArgs a) {
  if (a.gate && a.gate[0] == 1.f) return;
  segreduce_chunk_body<VEC, CH, MODE, true, true>(a, blockIdx.x);
}

// Partials of the run that starts in `piece` and spans np pieces, pieces q0, q0 + stride, ... (piece 0 of a run
// holds its sum on side 1, the others on side 0).  While whole rounds of U pieces remain, U partials are
// fetched together (independent loads in flight) and then added in the order of the walk; the rest goes one
// piece at a time.
template <int VEC, int CH>
__device__ __forceinline__ void seg_sum_partials(const SegArgs& a, int64_t piece, int np, int q0, int stride, int lg, bool keep2,
                                                 Acc<VEC, CH>& g, float& g2) {
  const int group = a.group, dim = a.dim;
  constexpr int U = CH == 1 ? 8 : (CH == 2 ? 4 : 2);
  int q = q0;
  for (; q + (U - 1) * stride < np; q += stride * U) {
    RxPack<VEC> v[U][CH];
    float p2[U];
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const int qj = q + j * stride;
      const int64_t e = (piece + qj) * 2 + (qj == 0 ? 1 : 0);
      const float* src = a.partials + e * (int64_t)dim;
#pragma unroll
      for (int k = 0; k < CH; ++k) {
        const int c = (lg + k * group) * VEC;
        v[j][k] = c < dim ? rx_load<VEC>(src + c) : rx_zero<VEC>();
      }
      p2[j] = (keep2 && lg == 0) ? a.partials2[e] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < U; ++j) {
#pragma unroll
      for (int k = 0; k < CH; ++k)
#pragma unroll
        for (int i = 0; i < VEC; ++i) g.p[k].v[i] += v[j][k].v[i];
      g2 += p2[j];
    }
  }
  for (; q < np; q += stride) {
    const int64_t e = (piece + q) * 2 + (q == 0 ? 1 : 0);
    const float* src = a.partials + e * (int64_t)dim;
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int c = (lg + k * group) * VEC;
      if (c < dim) {
        RxPack<VEC> v = rx_load<VEC>(src + c);
#pragma unroll
        for (int i = 0; i < VEC; ++i) g.p[k].v[i] += v.v[i];
      }
    }
    if (keep2 && lg == 0) g2 += a.partials2[e];
  }
}

#define SEG_LONG_RUN 64        // pieces: a run beyond this is added by the whole block, not by one wavefront
#define SEG_LONG_MAX_DIM 256   // widest row the block-wide path stages in LDS

template <int VEC, int CH, int MODE, bool FM>
__device__ __forceinline__ void segreduce_combine_body(const SegArgs& a, const int64_t n_pieces, const unsigned blk) {
  // A wavefront looks after 64/group consecutive pieces.  Each lane group first fetches what its
  // own piece's tail run needs (length, row, the row's current value), all groups at once; the
  // wavefront then takes the tails one by one, all lane slots adding that run's partials.
  // A hot row's run spans hundreds of pieces; walked by one wavefront it is one chain of dependent rounds
  // that sets the duration of the whole launch.  Such runs are set aside and added by all four wavefronts of
  // the block (every wavefront a quarter of the pieces, joined through LDS in wavefront order), after the
  // short ones.
  __shared__ float sq_lds[RX_BLOCK / 64];
  __shared__ int long_n;
  __shared__ int long_np[RX_BLOCK / 4], long_owner[RX_BLOCK / 4];
  __shared__ int64_t long_piece[RX_BLOCK / 4];
  __shared__ float long_part[RX_BLOCK / 64][SEG_LONG_MAX_DIM], long_part2[RX_BLOCK / 64];
  const int group = a.group, dim = a.dim;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slots = 64 / group;
  const int s = lane / group, lg = lane % group;
  const int64_t base = ((int64_t)blk * (RX_BLOCK / 64) + wave) * slots;
  const int64_t mine = base + s;
  const bool has2 = a.target2 != nullptr;
  const bool keep2 = FM || has2;
  const float scale = a.scale ? a.scale[0] : 1.f;
  const bool wide_ok = dim <= SEG_LONG_MAX_DIM;
  if (threadIdx.x == 0) long_n = 0;
  __syncthreads();
  const int u = mine < n_pieces ? a.tail_row[mine] : -1;
  int n_parts = 0;
  int64_t row = -1;
  if (u >= 0) {
    n_parts = (int)((a.seg_off[u + 1] - 1) / SEG_CHUNK - mine) + 1;
    row = a.uniq_rows[u];
  }
  const bool row_ok = row >= 0 && row < a.n_rows;
  Acc<VEC, CH> x, st;
  ACC_FOREACH(k, i) { x.p[k].v[i] = 0.f; st.p[k].v[i] = 0.f; }
  load_old<VEC, CH, MODE, FM>(a, row_ok ? row : 0, lg, x, st);
  float sq = 0.f;
  unsigned long long todo = __ballot(u >= 0 && lg == 0);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;  // wave-uniform
    todo &= todo - 1;
    const int owner = leader / group;
    const int np = __shfl(n_parts, leader, 64);
    const int64_t piece = base + owner;
    if (np > SEG_LONG_RUN && wide_ok) {
      if (lane == leader) {
        const int e = atomicAdd(&long_n, 1);      // (the order of the list does not enter any sum)
        long_np[e] = np; long_owner[e] = (int)threadIdx.x; long_piece[e] = piece;
      }
      continue;
    }
    Acc<VEC, CH> g;
    ACC_FOREACH(k, i) g.p[k].v[i] = 0.f;
    float g2 = 0.f;
    seg_sum_partials<VEC, CH>(a, piece, np, s, slots, lg, keep2, g, g2);
    for (int m = group; m < 64; m <<= 1) {
      ACC_FOREACH(k, i) g.p[k].v[i] += __shfl_xor(g.p[k].v[i], m, 64);
      g2 += __shfl_xor(g2, m, 64);
    }
    if (FM) g2 = __shfl(g2, lane - lg, 64);  // every lane of a group needs the scalar sum
    if (s == owner && row_ok) sq += finish_run<VEC, CH, MODE, FM>(a, g, g2, x, st, row, u, lg, scale, has2);
  }
  __syncthreads();
  const int nl = long_n;
  for (int e = 0; e < nl; ++e) {
    const int np = long_np[e], owner_tid = long_owner[e];
    const int64_t piece = long_piece[e];
    Acc<VEC, CH> g;
    ACC_FOREACH(k, i) g.p[k].v[i] = 0.f;
    float g2 = 0.f;
    seg_sum_partials<VEC, CH>(a, piece, np, wave * slots + s, (RX_BLOCK / 64) * slots, lg, keep2, g, g2);
    for (int m = group; m < 64; m <<= 1) {
      ACC_FOREACH(k, i) g.p[k].v[i] += __shfl_xor(g.p[k].v[i], m, 64);
      g2 += __shfl_xor(g2, m, 64);
    }
    if (s == 0) {
#pragma unroll
      for (int k = 0; k < CH; ++k) {
        const int c = (lg + k * group) * VEC;
        if (c < dim) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) long_part[wave][c + i] = g.p[k].v[i];
        }
      }
      if (lg == 0) long_part2[wave] = g2;
    }
    __syncthreads();
    if ((int)threadIdx.x - lg == owner_tid) {       // the lane group that holds the run's row and old value
      ACC_FOREACH(k, i) g.p[k].v[i] = 0.f;
#pragma unroll
      for (int k = 0; k < CH; ++k) {
        const int c = (lg + k * group) * VEC;
        if (c < dim) {
#pragma unroll
          for (int i = 0; i < VEC; ++i)
            g.p[k].v[i] = (long_part[0][c + i] + long_part[1][c + i]) + (long_part[2][c + i] + long_part[3][c + i]);
        }
      }
      g2 = (long_part2[0] + long_part2[1]) + (long_part2[2] + long_part2[3]);
      if (row_ok) sq += finish_run<VEC, CH, MODE, FM>(a, g, g2, x, st, row, u, lg, scale, has2);
    }
    __syncthreads();
  }
  if (a.sq_part) block_sq_store(sq, a.sq_part, sq_lds, blk);
}

template <int VEC, int CH, int MODE, bool FM>
__global__ void __launch_bounds__(RX_BLOCK)
segreduce_combine_kernel(SegArgs a, int64_t n_pieces) {
  if (a.gate && a.gate[0] == 1.f) return;
  segreduce_combine_body<VEC, CH, MODE, FM>(a, n_pieces, blockIdx.x);
}

template <int VEC, int CH, int MODE, bool FM>
__global__ void __launch_bounds__(RX_BLOCK)
segreduce_combine_multi_kernel(SegMulti m) {
  const int t = seg_multi_table(m);
  segreduce_combine_body<VEC, CH, MODE, false>(m.t[t], m.pieces[t], blockIdx.x - m.first[t]);
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
  RxRowShape sh = rx_row_shape(dim);
  if (sh.chunks == 3) sh.chunks = 4;
  return sh.group < 2 ? 2 : sh.group;
}

static int64_t seg_pieces(int64_t n) { return rx_cdiv(n, SEG_CHUNK); }
static int64_t seg_grid1(int64_t n, int dim) { return rx_cdiv(seg_pieces(n) * seg_group(dim), RX_BLOCK); }
static int64_t seg_grid2(int64_t n, int dim) { return rx_cdiv(rx_cdiv(seg_pieces(n), 64 / seg_group(dim)), RX_BLOCK / 64); }

extern "C" int64_t rx_segreduce_grid(int64_t n, int32_t dim) {
  if (n <= 0 || dim < 1 || dim > RX_MAX_DIM) return 0;
  return seg_grid1(n, dim) + seg_grid2(n, dim);
}

extern "C" size_t rx_segreduce_workspace_bytes(int64_t n, int32_t dim) {
  if (n <= 0 || dim < 1) return 256;
  const size_t pieces = (size_t)seg_pieces(n);
  return rx_align_up(pieces * 2 * dim * sizeof(float), 256) + rx_align_up(pieces * 2 * sizeof(float), 256) +
         rx_align_up(pieces * sizeof(int32_t), 256);
}

#define SEG_DISPATCH_MF(M, FMV, sh, KERNEL, grid, lds, st, ...)                                    \
  do {                                                                                             \
    if ((sh).vec == 4 && (sh).chunks == 1) hipLaunchKernelGGL((KERNEL<4, 1, M, FMV>), grid, dim3(RX_BLOCK), lds, st, __VA_ARGS__); \
    else if ((sh).vec == 4 && (sh).chunks == 2) hipLaunchKernelGGL((KERNEL<4, 2, M, FMV>), grid, dim3(RX_BLOCK), lds, st, __VA_ARGS__); \
    else if ((sh).vec == 4) hipLaunchKernelGGL((KERNEL<4, 4, M, FMV>), grid, dim3(RX_BLOCK), lds, st, __VA_ARGS__); \
    else if ((sh).chunks == 1) hipLaunchKernelGGL((KERNEL<1, 1, M, FMV>), grid, dim3(RX_BLOCK), lds, st, __VA_ARGS__); \
    else if ((sh).chunks == 2) hipLaunchKernelGGL((KERNEL<1, 2, M, FMV>), grid, dim3(RX_BLOCK), lds, st, __VA_ARGS__); \
    else hipLaunchKernelGGL((KERNEL<1, 4, M, FMV>), grid, dim3(RX_BLOCK), lds, st, __VA_ARGS__);  \
  } while (0)

#define SEG_DISPATCH_F(mode, FMV, sh, KERNEL, grid, lds, st, ...)                                  \
  do {                                                                                             \
    switch (mode) {                                                                                \
      case RX_SEG_DENSE: SEG_DISPATCH_MF(RX_SEG_DENSE, FMV, sh, KERNEL, grid, lds, st, __VA_ARGS__); break; \
      case RX_SEG_COMPACT: SEG_DISPATCH_MF(RX_SEG_COMPACT, FMV, sh, KERNEL, grid, lds, st, __VA_ARGS__); break; \
      case RX_SEG_SGD: SEG_DISPATCH_MF(RX_SEG_SGD, FMV, sh, KERNEL, grid, lds, st, __VA_ARGS__); break; \
      case RX_SEG_ADAGRAD: SEG_DISPATCH_MF(RX_SEG_ADAGRAD, FMV, sh, KERNEL, grid, lds, st, __VA_ARGS__); break; \
      case RX_SEG_SGD_SPEC: SEG_DISPATCH_MF(RX_SEG_SGD_SPEC, FMV, sh, KERNEL, grid, lds, st, __VA_ARGS__); break; \
      default: SEG_DISPATCH_MF(RX_SEG_ROWWISE_ADAGRAD, FMV, sh, KERNEL, grid, lds, st, __VA_ARGS__); break; \
    }                                                                                              \
  } while (0)

#define SEG_DISPATCH(mode, fm, sh, KERNEL, grid, lds, st, ...)                                     \
  do {                                                                                             \
    if (fm) SEG_DISPATCH_F(mode, true, sh, KERNEL, grid, lds, st, __VA_ARGS__);                    \
    else SEG_DISPATCH_F(mode, false, sh, KERNEL, grid, lds, st, __VA_ARGS__);                      \
  } while (0)

static int mode_check(int mode, const float* state, const char* who, bool allow_grad_modes) {
  RX_REQUIRE(mode >= (allow_grad_modes ? RX_SEG_DENSE : RX_SEG_SGD) && mode <= RX_SEG_SGD_SPEC, "%s: unknown mode %d", who, mode);
  RX_REQUIRE(!(mode == RX_SEG_ADAGRAD || mode == RX_SEG_ROWWISE_ADAGRAD) || state, "%s: Adagrad needs a state buffer", who);
  RX_REQUIRE(mode != RX_SEG_SGD_SPEC || state, "%s: the speculative step keeps its undo records in the state argument", who);
  return 0;
}

// Checks one table's arguments and places its cross-piece workspace at ws (a.n > 0).
static int seg_place(SegArgs& a, bool fm, const char* who, void* ws, size_t ws_bytes, RxRowShape* shp) {
  const int64_t n = a.n;
  const int dim = a.dim;
  RX_REQUIRE(n >= 0 && dim >= 1 && dim <= RX_MAX_DIM && a.occ_ld >= dim, "%s: bad shape n=%lld dim=%d ld=%lld", who, (long long)n, dim, (long long)a.occ_ld);
  if (mode_check(a.mode, a.state, who, true)) return 1;
  RX_REQUIRE(fm || (a.occ2 == nullptr) == (a.target2 == nullptr), "%s: companion gradient and target come as a pair", who);
  RX_REQUIRE(!a.target2 || !(a.mode == RX_SEG_ADAGRAD || a.mode == RX_SEG_ROWWISE_ADAGRAD || a.mode == RX_SEG_SGD_SPEC) || a.state2, "%s: companion Adagrad state / undo records missing", who);
  RX_REQUIRE(a.mode != RX_SEG_SGD_SPEC || !a.state2 || ((uintptr_t)a.state2 & 7) == 0, "%s: companion undo records must be 8-byte aligned", who);
  if (n == 0) return 0;
  RX_REQUIRE(a.occ && a.sorted_pos && a.sorted_slot && a.uniq_rows && a.seg_off && a.n_uniq && a.target && ws, "%s: null argument", who);
  RX_REQUIRE(a.n_rows >= 1, "%s: n_rows must be positive", who);
  RX_REQUIRE(ws_bytes >= rx_segreduce_workspace_bytes(n, dim) && rx_aligned16(ws), "%s: workspace too small or unaligned", who);
  if (seg_shape(dim, a.occ_ld, a.occ, a.target, (a.mode == RX_SEG_ADAGRAD || a.mode == RX_SEG_SGD_SPEC) ? a.state : nullptr, shp, who)) return 1;
  RX_REQUIRE(!fm || shp->vec == 1 || !a.fm_table || rx_aligned16(a.fm_table), "%s: table must be 16-byte aligned", who);
  const size_t pieces = (size_t)seg_pieces(n);
  char* w = (char*)ws;
  a.partials = (float*)w; w += rx_align_up(pieces * 2 * dim * sizeof(float), 256);
  a.partials2 = (float*)w; w += rx_align_up(pieces * 2 * sizeof(float), 256);
  a.tail_row = (int32_t*)w;
  a.group = shp->group;
  return 0;
}

// Shared driver of rx_segreduce and rx_segreduce_fm.
static int segreduce_run(SegArgs a, bool fm, const char* who, void* ws, size_t ws_bytes, hipStream_t st) {
  if (a.mode == RX_SEG_SGD_SPEC_OLD) {
    a.mode = RX_SEG_SGD_SPEC;
    a.undo_old_only = 1;
  } else if (a.mode == RX_SEG_COMPACT_IF_CLIPPED) {
    RX_REQUIRE(a.scale, "%s: RX_SEG_COMPACT_IF_CLIPPED needs the clip factor", who);
    a.mode = RX_SEG_COMPACT;
    a.gate = a.scale;
  }
  RxRowShape sh;
  if (seg_place(a, fm, who, ws, ws_bytes, &sh)) return 1;
  const int64_t n = a.n;
  const int dim = a.dim;
  if (n == 0) return 0;
  const size_t pieces = (size_t)seg_pieces(n);
  const unsigned g1 = (unsigned)seg_grid1(n, dim), g2 = (unsigned)seg_grid2(n, dim);
  const size_t tile = (size_t)(RX_BLOCK / sh.group) * (SEG_CHUNK + 1);
  const bool deep = fm && a.fm_deep != nullptr;
  const size_t lds = (tile + 2) * sizeof(int2) + tile * sizeof(uint32_t) * (deep ? 2 : 1);
  if (deep) SEG_DISPATCH_F(a.mode, true, sh, segreduce_chunk_deep_kernel, dim3(g1), lds, st, a);
  else SEG_DISPATCH(a.mode, fm, sh, segreduce_chunk_kernel, dim3(g1), lds, st, a);
  RX_LAUNCH_CHECK();
  if (a.sq_part) a.sq_part = a.sq_part + g1;
  SEG_DISPATCH(a.mode, fm, sh, segreduce_combine_kernel, dim3(g2), 0, st, a, (int64_t)pieces);
  RX_LAUNCH_CHECK();
  return 0;
}

static void seg_from_table(const rx_seg_table& t, int mode, float lr, float eps, const float* d_scale, SegArgs* out) {
  SegArgs a{};
  a.occ = t.occ; a.occ_ld = t.occ_ld; a.n = t.n; a.dim = t.dim;
  a.sorted_pos = t.sorted_pos; a.sorted_slot = t.sorted_slot; a.uniq_rows = t.uniq_rows;
  a.seg_off = t.seg_off; a.n_uniq = t.n_uniq; a.mode = mode; a.target = t.target; a.state = t.state;
  a.n_rows = t.n_rows; a.lr = lr; a.eps = eps; a.scale = d_scale; a.sq_part = t.sq_part;
  a.occ2 = t.occ2; a.target2 = t.target2; a.state2 = t.state2;
  *out = a;
}

extern "C" int rx_segreduce_multi(const rx_seg_table* tables, int32_t count, int mode, float lr, float eps,
                                  const float* d_scale, void* ws, size_t ws_bytes, int device, void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(count >= 0 && count <= SEG_MULTI_MAX && (tables || count == 0), "rx_segreduce_multi: 0..%d tables per call", SEG_MULTI_MAX);
  RX_REQUIRE(mode != RX_SEG_SGD_SPEC, "rx_segreduce_multi: the speculative step goes through rx_segreduce / rx_segreduce_fm");
  static_assert(SEG_MULTI_MAX == RX_SEG_MULTI_MAX, "header and kernel disagree on the table limit");
  SegMulti m1{}, m2{};
  RxRowShape sh0{};
  char* w = (char*)ws;
  size_t left = ws_bytes, lds = 0;
  unsigned b1 = 0, b2 = 0;
  int k = 0;
  for (int i = 0; i < count; ++i) {
    SegArgs a;
    seg_from_table(tables[i], mode, lr, eps, d_scale, &a);
    RxRowShape sh;
    if (seg_place(a, false, "rx_segreduce_multi", w, left, &sh)) return 1;
    if (a.n == 0) continue;
    RX_REQUIRE(k == 0 || (sh.vec == sh0.vec && sh.chunks == sh0.chunks),
               "rx_segreduce_multi: the tables of one call must share the row shape (widths %d and %d do not)", m1.t[0].dim, a.dim);
    for (int j = 0; j < k; ++j)
      RX_REQUIRE(m1.t[j].target != a.target, "rx_segreduce_multi: a target may appear once per call (two writers of one row would race)");
    sh0 = sh;
    const size_t need = rx_segreduce_workspace_bytes(a.n, a.dim);
    w += need; left -= need;
    const unsigned g1 = (unsigned)seg_grid1(a.n, a.dim), g2 = (unsigned)seg_grid2(a.n, a.dim);
    const size_t tile = (size_t)(RX_BLOCK / sh.group) * (SEG_CHUNK + 1);
    const size_t l = (tile + 2) * sizeof(int2) + tile * sizeof(uint32_t);
    lds = l > lds ? l : lds;
    m1.t[k] = a; m1.first[k] = b1; b1 += g1;
    if (a.sq_part) a.sq_part += g1;            // the combine pass writes its partials behind those of pass 1
    m2.t[k] = a; m2.first[k] = b2; m2.pieces[k] = seg_pieces(a.n); b2 += g2;
    ++k;
  }
  if (k == 0) return 0;
  m1.first[k] = b1; m2.first[k] = b2; m1.count = m2.count = k;
  hipStream_t st = (hipStream_t)stream;
  SEG_DISPATCH_F(mode, false, sh0, segreduce_chunk_multi_kernel, dim3(b1), lds, st, m1);
  RX_LAUNCH_CHECK();
  SEG_DISPATCH_F(mode, false, sh0, segreduce_combine_multi_kernel, dim3(b2), 0, st, m2);
  RX_LAUNCH_CHECK();
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
  SegArgs a{};
  a.occ = d_occ; a.occ_ld = occ_ld; a.n = n; a.dim = dim;
  a.sorted_pos = d_sorted_pos; a.sorted_slot = d_sorted_slot; a.uniq_rows = d_uniq_rows;
  a.seg_off = d_seg_off; a.n_uniq = d_n_uniq; a.mode = mode; a.target = d_target; a.state = d_state;
  a.n_rows = n_rows; a.lr = lr; a.eps = eps; a.scale = d_scale; a.sq_part = d_sq_part;
  a.occ2 = d_occ2; a.target2 = d_target2; a.state2 = d_state2;
  return segreduce_run(a, false, "rx_segreduce", ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int rx_segreduce_fm(const float* d_T, int64_t ld_T, const float* d_g, const float* d_deep,
                               int64_t deep_ld, int64_t n_samples, int32_t F, int32_t dim,
                               const int32_t* d_sorted_pos,
                               const int32_t* d_sorted_slot, const int64_t* d_uniq_rows,
                               const int32_t* d_seg_off, const int32_t* d_n_uniq, int mode,
                               float* d_target, float* d_state, const float* d_table, float* d_target2,
                               float* d_state2, int64_t n_rows, float lr, float eps, const float* d_scale,
                               float* d_sq_part, void* ws, size_t ws_bytes, int device, void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(n_samples >= 0 && F >= 1, "rx_segreduce_fm: bad shape");
  RX_REQUIRE(d_g || n_samples == 0, "rx_segreduce_fm: null logit gradient");
  RX_REQUIRE(n_samples * (int64_t)(F + 1) < (1ll << 31), "rx_segreduce_fm: too many occurrences");
  const bool grad_mode = mode == RX_SEG_DENSE || mode == RX_SEG_COMPACT;
  RX_REQUIRE(!grad_mode || d_table, "rx_segreduce_fm: gradient modes need the table to read rows from");
  SegArgs a{};
  a.occ = d_T; a.occ_ld = ld_T; a.n = n_samples * (int64_t)(F + 1); a.dim = dim;
  a.sorted_pos = d_sorted_pos; a.sorted_slot = d_sorted_slot; a.uniq_rows = d_uniq_rows;
  a.seg_off = d_seg_off; a.n_uniq = d_n_uniq; a.mode = mode; a.target = d_target; a.state = d_state;
  a.n_rows = n_rows; a.lr = lr; a.eps = eps; a.scale = d_scale; a.sq_part = d_sq_part;
  a.occ2 = nullptr; a.target2 = d_target2; a.state2 = d_state2;
  a.fm_g = d_g; a.fm_table = d_table; a.fm_split = n_samples * (int64_t)F; a.fm_div = F;
  if (d_deep) {
    RX_REQUIRE(deep_ld >= (int64_t)(F + 1) * dim && n_samples * deep_ld < (1ll << 32), "rx_segreduce_fm: bad tower-gradient stride");
    RX_REQUIRE(dim % 4 != 0 || (deep_ld % 4 == 0 && rx_aligned16(d_deep)), "rx_segreduce_fm: tower gradient must be 16-byte aligned with a row stride that is a multiple of 4");
    a.fm_deep = d_deep; a.fm_deep_ld = deep_ld;
  }
  return segreduce_run(a, true, "rx_segreduce_fm", ws, ws_bytes, (hipStream_t)stream);
}

// ---------------------------------------------------------------- update from a compact gradient
template <int CH> __host__ __device__ constexpr int upd_rows() { return CH == 1 ? 4 : (CH == 2 ? 2 : 1); }
static inline int upd_rows_of(int chunks, int mode) { return mode == RX_SEG_SGD_SPEC ? 1 : (chunks == 1 ? 4 : (chunks == 2 ? 2 : 1)); }

template <int VEC, int CH, int MODE>
__device__ __forceinline__ void row_update_body(const SegArgs& a, const unsigned blk, const float scale) {
  const int group = a.group, dim = a.dim;
  const int64_t gid = (int64_t)blk * blockDim.x + threadIdx.x;
  const int lg = (int)(gid % group);
  const int64_t u = (gid / group) * (MODE == RX_SEG_SGD_SPEC ? 1 : upd_rows<CH>());
  if (u >= a.n_uniq[0]) return;
  if constexpr (MODE == RX_SEG_SGD_SPEC) {
    const int64_t row = a.uniq_rows[u];
    if (row < 0 || row >= a.n_rows) return;
    float* w = a.target + row * (int64_t)dim;
    if (a.undo_old_only) {      // records of what was there before the step: put it back
      const float* ud = a.occ + u * (int64_t)dim;
#pragma unroll
      for (int k = 0; k < CH; ++k) {
        const int c = (lg + k * group) * VEC;
        if (c < dim) rx_store<VEC>(w + c, rx_load<VEC>(ud + c));
      }
      if (a.target2 && lg == 0) a.target2[row] = a.occ2[u];
      return;
    }
    // redo the row from its record: old - lr * (factor * G), what RX_SEG_COMPACT + RX_SEG_SGD leave
    const float* ud = a.occ + u * (int64_t)(2 * dim);
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int c = (lg + k * group) * VEC;
      if (c < dim) {
        RxPack<VEC> G = rx_load<VEC>(ud + c), x = rx_load<VEC>(ud + dim + c);
#pragma unroll
        for (int i = 0; i < VEC; ++i) x.v[i] = x.v[i] - a.lr * (G.v[i] * scale);
        rx_store<VEC>(w + c, x);
      }
    }
    if (a.target2 && lg == 0) {
      const float2 t = reinterpret_cast<const float2*>(a.occ2)[u];
      a.target2[row] = t.y - a.lr * (t.x * scale);
    }
    return;
  }
  // UPD_ROWS rows per lane group, every load of all of them issued before the first use: a row costs three
  // dependent round trips (count, row id, row), and one row per group left the launch bound by that latency
  // (4.0 TB/s on 512-byte rows) rather than by the bytes.
  constexpr int R = upd_rows<CH>();
  const int64_t nu = a.n_uniq[0];
  int64_t rows[R];
  bool ok[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int64_t ur = u + r;
    ok[r] = ur < nu;
    rows[r] = a.uniq_rows[ok[r] ? ur : 0];
    ok[r] = ok[r] && rows[r] >= 0 && rows[r] < a.n_rows;
  }
  Acc<VEC, CH> g[R], x[R], st[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int64_t ur = ok[r] ? u + r : 0;
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int c = (lg + k * group) * VEC;
      g[r].p[k] = (c < dim) ? rx_load<VEC>(a.occ + ur * (int64_t)dim + c) : rx_zero<VEC>();
    }
    ACC_FOREACH(k, i) { x[r].p[k].v[i] = 0.f; st[r].p[k].v[i] = 0.f; }
    load_old<VEC, CH, MODE, false>(a, ok[r] ? rows[r] : 0, lg, x[r], st[r]);
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (!ok[r]) continue;
    apply_row<VEC, CH, MODE>(a, g[r], x[r], st[r], rows[r], u + r, lg, scale);
    // companion scalar table addressed by the same rows
    if (a.target2 && lg == 0) apply_scalar<MODE>(a, a.occ2[u + r], rows[r], u + r, scale);
  }
}

template <int VEC, int CH, int MODE, bool FM>
__global__ void __launch_bounds__(RX_BLOCK)
row_update_kernel(SegArgs a) {
  if constexpr (MODE == RX_SEG_SGD_SPEC) {
    // the reduce already took the step with a factor of 1: nothing to do unless the clip bites.  A small grid
    // strides over the list, so that the common case costs a few hundred blocks reading one float.
    const float scale = a.scale ? a.scale[0] : 1.f;
    if (scale == 1.f && !a.force) return;
    for (unsigned b = blockIdx.x; b < a.upd_blocks; b += gridDim.x) row_update_body<VEC, CH, MODE>(a, b, scale);
  } else {
    if (a.gate && a.gate[0] == 1.f) return;
    row_update_body<VEC, CH, MODE>(a, blockIdx.x, a.scale ? a.scale[0] : 1.f);
  }
}

template <int VEC, int CH, int MODE, bool FM>
__global__ void __launch_bounds__(RX_BLOCK)
row_update_multi_kernel(SegMulti m) {
  const int t = seg_multi_table(m);
  row_update_body<VEC, CH, MODE>(m.t[t], blockIdx.x - m.first[t], m.t[t].scale ? m.t[t].scale[0] : 1.f);
}

extern "C" int rx_row_update(const float* d_compact, int32_t dim, const int64_t* d_uniq_rows,
                             const int32_t* d_n_uniq, int64_t n_max, int mode, float* d_w,
                             float* d_state, const float* d_compact2, float* d_w2, float* d_state2,
                             int64_t n_rows, float lr, float eps, const float* d_scale, int device,
                             void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(n_max >= 0 && dim >= 1 && dim <= RX_MAX_DIM, "rx_row_update: bad shape");
  const bool force = mode == RX_SEG_SGD_SPEC_REDO || mode == RX_SEG_RESTORE;
  const bool restore = mode == RX_SEG_RESTORE || mode == RX_SEG_RESTORE_IF_CLIPPED;
  const bool gated = mode == RX_SEG_SGD_IF_CLIPPED;
  RX_REQUIRE(!(mode == RX_SEG_RESTORE_IF_CLIPPED || gated) || d_scale, "rx_row_update: the *_IF_CLIPPED modes need the clip factor");
  if (gated) mode = RX_SEG_SGD;
  if (force || restore) mode = RX_SEG_SGD_SPEC;
  if (mode == RX_SEG_SGD_SPEC) {      // d_compact / d_compact2 are the undo records; no optimizer state
    RX_REQUIRE(d_scale || (restore && force), "rx_row_update: the speculative step needs the clip factor");
    d_state = const_cast<float*>(d_compact);
  }
  if (mode_check(mode, d_state, "rx_row_update", false)) return 1;
  RX_REQUIRE((d_compact2 == nullptr) == (d_w2 == nullptr), "rx_row_update: companion gradient and table come as a pair");
  RX_REQUIRE(!d_w2 || !(mode == RX_SEG_ADAGRAD || mode == RX_SEG_ROWWISE_ADAGRAD) || d_state2, "rx_row_update: companion Adagrad state missing");
  if (n_max == 0) return 0;
  RX_REQUIRE(d_compact && d_uniq_rows && d_n_uniq && d_w && n_rows >= 1, "rx_row_update: null argument");
  RxRowShape sh;
  if (seg_shape(dim, dim, d_compact, d_w, mode == RX_SEG_ADAGRAD ? d_state : nullptr, &sh, "rx_row_update")) return 1;
  SegArgs a{};
  a.occ = d_compact; a.occ_ld = dim; a.n = n_max; a.dim = dim; a.uniq_rows = d_uniq_rows; a.n_uniq = d_n_uniq;
  a.mode = mode; a.target = d_w; a.state = d_state; a.n_rows = n_rows; a.lr = lr; a.eps = eps; a.scale = d_scale;
  a.occ2 = d_compact2; a.target2 = d_w2; a.state2 = d_state2;
  a.group = sh.group;
  unsigned grid = (unsigned)rx_cdiv(rx_cdiv(n_max, upd_rows_of(sh.chunks, mode)) * sh.group, RX_BLOCK);
  a.force = force ? 1 : 0;
  a.undo_old_only = restore ? 1 : 0;
  a.gate = gated ? d_scale : nullptr;
  a.upd_blocks = grid;
  if (mode == RX_SEG_SGD_SPEC && grid > 2048u) grid = 2048u;
  SEG_DISPATCH_F(mode, false, sh, row_update_kernel, dim3(grid), 0, (hipStream_t)stream, a);
  RX_LAUNCH_CHECK();
  return 0;
}

extern "C" int rx_row_update_multi(const rx_seg_table* tables, int32_t count, int mode, float lr, float eps,
                                   const float* d_scale, int device, void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(count >= 0 && count <= SEG_MULTI_MAX && (tables || count == 0), "rx_row_update_multi: 0..%d tables per call", SEG_MULTI_MAX);
  RX_REQUIRE(mode != RX_SEG_SGD_SPEC, "rx_row_update_multi: the speculative step goes through rx_row_update");
  SegMulti m{};
  RxRowShape sh0{};
  unsigned blocks = 0;
  int k = 0;
  for (int i = 0; i < count; ++i) {
    const rx_seg_table& t = tables[i];
    RX_REQUIRE(t.n >= 0 && t.dim >= 1 && t.dim <= RX_MAX_DIM, "rx_row_update_multi: bad shape");
    if (mode_check(mode, t.state, "rx_row_update_multi", false)) return 1;
    RX_REQUIRE((t.occ2 == nullptr) == (t.target2 == nullptr), "rx_row_update_multi: companion gradient and table come as a pair");
    RX_REQUIRE(!t.target2 || !(mode == RX_SEG_ADAGRAD || mode == RX_SEG_ROWWISE_ADAGRAD) || t.state2, "rx_row_update_multi: companion Adagrad state missing");
    if (t.n == 0) continue;
    RX_REQUIRE(t.occ && t.uniq_rows && t.n_uniq && t.target && t.n_rows >= 1, "rx_row_update_multi: null argument");
    for (int j = 0; j < k; ++j)
      RX_REQUIRE(m.t[j].target != t.target, "rx_row_update_multi: a table may appear once per call (two updates of one row would race)");
    RxRowShape sh;
    if (seg_shape(t.dim, t.dim, t.occ, t.target, mode == RX_SEG_ADAGRAD ? t.state : nullptr, &sh, "rx_row_update_multi")) return 1;
    RX_REQUIRE(k == 0 || (sh.vec == sh0.vec && sh.chunks == sh0.chunks), "rx_row_update_multi: the tables of one call must share the row shape");
    sh0 = sh;
    SegArgs a;
    seg_from_table(t, mode, lr, eps, d_scale, &a);
    a.occ_ld = t.dim; a.group = sh.group; a.sq_part = nullptr;
    m.t[k] = a; m.first[k] = blocks;
    blocks += (unsigned)rx_cdiv(rx_cdiv(t.n, upd_rows_of(sh.chunks, mode)) * sh.group, RX_BLOCK);
    ++k;
  }
  if (k == 0) return 0;
  m.first[k] = blocks; m.count = k;
  SEG_DISPATCH_F(mode, false, sh0, row_update_multi_kernel, dim3(blocks), 0, (hipStream_t)stream, m);
  RX_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- dense optimizers
static inline unsigned dense_grid(int64_t n) {
  int64_t g = rx_cdiv(n, RX_BLOCK);
  return (unsigned)(g < 2048 ? g : 2048);
}

// The step count t enters through the bias corrections 1 - beta^t only.  It comes either by value or
// from device memory (a counter the caller advances on the stream, so that a recorded graph of the
// step stays valid from one replay to the next); both forms run the same arithmetic in the kernel.
__global__ void __launch_bounds__(RX_BLOCK)
dense_adam_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m,
                  float* __restrict__ v, int64_t n, float lr, float beta1, float beta2, float eps,
                  int32_t step, const int32_t* __restrict__ d_step, const float* __restrict__ scale) {
  const float sc = scale ? scale[0] : 1.f;
  const int32_t t = d_step ? d_step[0] : step;
  if (t < 1) return;                                   // a counter that was never advanced: leave the weights alone
  const double bc1 = 1.0 - pow((double)beta1, (double)t);
  const double bc2 = 1.0 - pow((double)beta2, (double)t);
  const float step_size = (float)((double)lr / bc1);
  const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
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

static int dense_adam_run(float* d_w, const float* d_g, float* d_m, float* d_v, int64_t n, float lr, float beta1,
                          float beta2, float eps, int32_t step, const int32_t* d_step, const float* d_scale,
                          hipStream_t st) {
  if (n == 0) return 0;
  RX_REQUIRE(d_w && d_g && d_m && d_v, "rx_dense_adam: null argument");
  hipLaunchKernelGGL(dense_adam_kernel, dim3(dense_grid(n)), dim3(RX_BLOCK), 0, st, d_w, d_g, d_m, d_v, n, lr, beta1,
                     beta2, eps, step, d_step, d_scale);
  RX_LAUNCH_CHECK();
  return 0;
}

extern "C" int rx_dense_adam(float* d_w, const float* d_g, float* d_m, float* d_v, int64_t n, float lr,
                             float beta1, float beta2, float eps, int32_t step, const float* d_scale,
                             int device, void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(n >= 0 && step >= 1, "rx_dense_adam: bad n or step");
  return dense_adam_run(d_w, d_g, d_m, d_v, n, lr, beta1, beta2, eps, step, nullptr, d_scale, (hipStream_t)stream);
}

extern "C" int rx_dense_adam_dstep(float* d_w, const float* d_g, float* d_m, float* d_v, int64_t n, float lr,
                                   float beta1, float beta2, float eps, const int32_t* d_step,
                                   const float* d_scale, int device, void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(n >= 0 && d_step, "rx_dense_adam_dstep: bad n or null step counter");
  return dense_adam_run(d_w, d_g, d_m, d_v, n, lr, beta1, beta2, eps, 0, d_step, d_scale, (hipStream_t)stream);
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

// ---------------------------------------------------------------- many small dense tensors at once
// The towers have a dozen weight / bias tensors of a few KB each; one launch per tensor and
// reduction would cost more in launch latency than in work.  blockIdx.y selects the tensor.
#define RX_MULTI_BLOCKS 32

__global__ void __launch_bounds__(RX_BLOCK)
multi_sumsq_kernel(rx_tensor_list t, double* __restrict__ partial) {
  __shared__ double wsum[RX_BLOCK / 64];
  const int k = blockIdx.y;
  const float* __restrict__ x = t.g[k];
  const int64_t n = t.n[k];
  double acc = 0.0;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
  // 16-byte pieces, four of them in flight per thread (a 0.5 M-element weight matrix is 16 trips to memory
  // per thread instead of 64 dependent ones); the elements past the last whole piece one by one
  const int64_t n4 = (((uintptr_t)x & 15) == 0) ? n / 4 : 0;
  const float4* __restrict__ x4 = reinterpret_cast<const float4*>(x);
  int64_t i = tid;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    const float4 a = x4[i], b = x4[i + stride], c = x4[i + 2 * stride], d = x4[i + 3 * stride];
    acc += ((double)a.x * a.x + (double)a.y * a.y) + ((double)a.z * a.z + (double)a.w * a.w);
    acc += ((double)b.x * b.x + (double)b.y * b.y) + ((double)b.z * b.z + (double)b.w * b.w);
    acc += ((double)c.x * c.x + (double)c.y * c.y) + ((double)c.z * c.z + (double)c.w * c.w);
    acc += ((double)d.x * d.x + (double)d.y * d.y) + ((double)d.z * d.z + (double)d.w * d.w);
  }
  for (; i < n4; i += stride) {
    const float4 a = x4[i];
    acc += ((double)a.x * a.x + (double)a.y * a.y) + ((double)a.z * a.z + (double)a.w * a.w);
  }
  for (int64_t j = n4 * 4 + tid; j < n; j += stride) {
    const double v = (double)x[j];
    acc += v * v;
  }
  acc = rx_wave_sum_f64(acc);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[(int64_t)k * RX_MULTI_BLOCKS + blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

template <int MODE>
__device__ __forceinline__ void multi_dense_body(const rx_tensor_list& t, const int k, const unsigned bx, const unsigned nbx, float lr, float eps,
                                                 const float sc) {
  float* __restrict__ w = t.w[k];
  const float* __restrict__ g = t.g[k];
  float* __restrict__ s = t.state[k];
  const int64_t n = t.n[k];
  const int64_t tid = (int64_t)bx * blockDim.x + threadIdx.x, stride = (int64_t)nbx * blockDim.x;
  auto one = [&](float wi, float gi_, float& si) {   // the rule on one element (same arithmetic whatever the access width)
    const float gi = gi_ * sc;
    if (MODE == RX_SEG_SGD) return wi - lr * gi;
    si = si + gi * gi;
    return wi - lr * (gi / (sqrtf(si) + eps));
  };
  // 16-byte pieces, two in flight per thread, where all of the tensor's arrays allow it
  const bool vec = (((uintptr_t)w | (uintptr_t)g | (MODE == RX_SEG_SGD ? 0 : (uintptr_t)s)) & 15) == 0;
  const int64_t n4 = vec ? n / 4 : 0;
  float4* __restrict__ w4 = reinterpret_cast<float4*>(w);
  const float4* __restrict__ g4 = reinterpret_cast<const float4*>(g);
  float4* __restrict__ s4 = reinterpret_cast<float4*>(s);
  auto piece = [&](int64_t i, float4 wv, float4 gv, float4 sv) {
    wv.x = one(wv.x, gv.x, sv.x); wv.y = one(wv.y, gv.y, sv.y); wv.z = one(wv.z, gv.z, sv.z); wv.w = one(wv.w, gv.w, sv.w);
    w4[i] = wv;
    if (MODE != RX_SEG_SGD) s4[i] = sv;
  };
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  int64_t i = tid;
  for (; i + stride < n4; i += 2 * stride) {
    const float4 wa = w4[i], ga = g4[i], wb = w4[i + stride], gb = g4[i + stride];
    const float4 sa = MODE == RX_SEG_SGD ? z4 : s4[i], sb = MODE == RX_SEG_SGD ? z4 : s4[i + stride];
    piece(i, wa, ga, sa);
    piece(i + stride, wb, gb, sb);
  }
  for (; i < n4; i += stride) piece(i, w4[i], g4[i], MODE == RX_SEG_SGD ? z4 : s4[i]);
  for (int64_t j = n4 * 4 + tid; j < n; j += stride) {
    float sj = MODE == RX_SEG_SGD ? 0.f : s[j];
    w[j] = one(w[j], g[j], sj);
    if (MODE != RX_SEG_SGD) s[j] = sj;
  }
}

template <int MODE>
__global__ void __launch_bounds__(RX_BLOCK)
multi_dense_update_kernel(rx_tensor_list t, float lr, float eps, const float* __restrict__ scale) {
  multi_dense_body<MODE>(t, blockIdx.y, blockIdx.x, gridDim.x, lr, eps, scale ? scale[0] : 1.f);
}

// Adam on up to RX_MULTI_MAX tensors in one launch (blockIdx.y selects the tensor): the arithmetic of
// dense_adam_kernel element by element, so the weights equal those of one launch per tensor bit for bit.
#define RX_MULTI_ADAM_BLOCKS 64
__global__ void __launch_bounds__(RX_BLOCK)
multi_dense_adam_kernel(rx_adam_list t, float lr, float beta1, float beta2, float eps, int32_t step,
                        const int32_t* __restrict__ d_step, const float* __restrict__ scale) {
  const int k = blockIdx.y;
  float* __restrict__ w = t.w[k];
  const float* __restrict__ g = t.g[k];
  float* __restrict__ m = t.m[k];
  float* __restrict__ v = t.v[k];
  const int64_t n = t.n[k];
  const float sc = scale ? scale[0] : 1.f;
  const int32_t tt = d_step ? d_step[0] : step;
  if (tt < 1) return;
  const double bc1 = 1.0 - pow((double)beta1, (double)tt);
  const double bc2 = 1.0 - pow((double)beta2, (double)tt);
  const float step_size = (float)((double)lr / bc1);
  const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  auto one = [&](float& wi, float gi_, float& mi_, float& vi_) {
    const float gi = gi_ * sc;
    const float mi = mi_ + (gi - mi_) * (1.f - beta1);
    const float vi = vi_ * beta2 + (1.f - beta2) * gi * gi;
    mi_ = mi;
    vi_ = vi;
    const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
    wi = wi - step_size * (mi / denom);
  };
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
  const bool vec = (((uintptr_t)w | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0;
  const int64_t n4 = vec ? n / 4 : 0;
  float4* __restrict__ w4 = reinterpret_cast<float4*>(w);
  const float4* __restrict__ g4 = reinterpret_cast<const float4*>(g);
  float4* __restrict__ m4 = reinterpret_cast<float4*>(m);
  float4* __restrict__ v4 = reinterpret_cast<float4*>(v);
  for (int64_t i = tid; i < n4; i += stride) {
    float4 wv = w4[i], mv = m4[i], vv = v4[i];
    const float4 gv = g4[i];
    one(wv.x, gv.x, mv.x, vv.x); one(wv.y, gv.y, mv.y, vv.y); one(wv.z, gv.z, mv.z, vv.z); one(wv.w, gv.w, mv.w, vv.w);
    w4[i] = wv; m4[i] = mv; v4[i] = vv;
  }
  for (int64_t j = n4 * 4 + tid; j < n; j += stride) {
    float wj = w[j], mj = m[j], vj = v[j];
    one(wj, g[j], mj, vj);
    w[j] = wj; m[j] = mj; v[j] = vj;
  }
}

extern "C" int rx_multi_dense_adam(const rx_adam_list* list, float lr, float beta1, float beta2, float eps, int32_t step,
                                   const int32_t* d_step, const float* d_scale, int device, void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(list && list->count >= 0 && list->count <= RX_MULTI_MAX, "rx_multi_dense_adam: bad tensor list");
  RX_REQUIRE(d_step || step >= 1, "rx_multi_dense_adam: step count must be at least 1 (or live on the device)");
  if (list->count == 0) return 0;
  for (int k = 0; k < list->count; ++k)
    RX_REQUIRE(list->n[k] >= 0 && ((list->w[k] && list->g[k] && list->m[k] && list->v[k]) || list->n[k] == 0), "rx_multi_dense_adam: bad tensor %d", k);
  hipLaunchKernelGGL(multi_dense_adam_kernel, dim3(RX_MULTI_ADAM_BLOCKS, list->count), dim3(RX_BLOCK), 0, (hipStream_t)stream, *list, lr,
                     beta1, beta2, eps, step, d_step, d_scale);
  RX_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t rx_multi_sumsq_partials(int32_t count) {
  return (size_t)(count > 0 ? count : 0) * RX_MULTI_BLOCKS;
}

extern "C" int rx_multi_sumsq(const rx_tensor_list* list, double* d_partial, int device, void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(list && list->count >= 0 && list->count <= RX_MULTI_MAX, "rx_multi_sumsq: bad tensor list");
  if (list->count == 0) return 0;
  RX_REQUIRE(d_partial, "rx_multi_sumsq: null partial buffer");
  for (int k = 0; k < list->count; ++k) RX_REQUIRE(list->n[k] >= 0 && (list->g[k] || list->n[k] == 0), "rx_multi_sumsq: bad tensor %d", k);
  hipLaunchKernelGGL(multi_sumsq_kernel, dim3(RX_MULTI_BLOCKS, list->count), dim3(RX_BLOCK), 0, (hipStream_t)stream, *list, d_partial);
  RX_LAUNCH_CHECK();
  return 0;
}

extern "C" int rx_multi_dense_update(const rx_tensor_list* list, int mode, float lr, float eps, const float* d_scale,
                                     int device, void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(list && list->count >= 0 && list->count <= RX_MULTI_MAX, "rx_multi_dense_update: bad tensor list");
  RX_REQUIRE(mode == RX_SEG_SGD || mode == RX_SEG_ADAGRAD, "rx_multi_dense_update: mode must be SGD or ADAGRAD");
  if (list->count == 0) return 0;
  for (int k = 0; k < list->count; ++k)
    RX_REQUIRE(list->n[k] >= 0 && ((list->w[k] && list->g[k] && (mode == RX_SEG_SGD || list->state[k])) || list->n[k] == 0), "rx_multi_dense_update: bad tensor %d", k);
  const dim3 grid(RX_MULTI_BLOCKS, list->count);
  if (mode == RX_SEG_SGD) hipLaunchKernelGGL(multi_dense_update_kernel<RX_SEG_SGD>, grid, dim3(RX_BLOCK), 0, (hipStream_t)stream, *list, lr, eps, d_scale);
  else hipLaunchKernelGGL(multi_dense_update_kernel<RX_SEG_ADAGRAD>, grid, dim3(RX_BLOCK), 0, (hipStream_t)stream, *list, lr, eps, d_scale);
  RX_LAUNCH_CHECK();
  return 0;
}

// Thread t adds parts[t], parts[t + 256], ... in order (fp64), fixed xor trees and a fixed order over the four
// waves join them; then the clip factor.  Every thread of the block gets the factor (and the norm in *tn).
__device__ __forceinline__ float clip_parts_block(const float* __restrict__ parts, int64_t n, const double* __restrict__ parts64, int64_t n64,
                                                  const double* __restrict__ acc, float max_norm, double* wsum, float* tn_out) {
  double s = 0.0;
  // eight loads in flight, added in index order: the sum is the one of the plain loop, bit for bit
  int64_t i = threadIdx.x;
  for (; i + 7 * RX_BLOCK < n; i += 8 * RX_BLOCK) {
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = parts[i + k * RX_BLOCK];
#pragma unroll
    for (int k = 0; k < 8; ++k) s += (double)v[k];
  }
  for (; i < n; i += RX_BLOCK) s += (double)parts[i];
  for (int64_t j = threadIdx.x; j < n64; j += RX_BLOCK) s += parts64[j];
  s = rx_wave_sum_f64(s);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
  __syncthreads();
  const double total = (acc ? acc[0] : 0.0) + ((wsum[0] + wsum[1]) + (wsum[2] + wsum[3]));
  const float tn = (float)sqrt(total);
  const float coef = max_norm / (tn + 1e-6f);
  *tn_out = tn;
  return coef < 1.f ? coef : 1.f;
}

__global__ void __launch_bounds__(RX_BLOCK)
clip_scale_parts_kernel(const float* __restrict__ parts, int64_t n, const double* __restrict__ parts64, int64_t n64,
                        const double* __restrict__ acc, float max_norm, float* __restrict__ scale,
                        float* __restrict__ total_norm) {
  __shared__ double wsum[RX_BLOCK / 64];
  float tn;
  const float sc = clip_parts_block(parts, n, parts64, n64, acc, max_norm, wsum, &tn);
  if (threadIdx.x != 0) return;
  scale[0] = sc;
  if (total_norm) total_norm[0] = tn;
}

// The tail of a clipped step in ONE launch: clip factor (formed by every block from the step's partial sums of
// squares: same order and tree as clip_scale_parts_kernel, so all blocks hold the same bits), the row updates of up
// to RX_SEG_MULTI_MAX tables (blocks [0, row_blocks)) and the update of up to RX_MULTI_MAX dense tensors (the
// blocks behind them, RX_MULTI_BLOCKS per tensor).  Results equal rx_clip_scale_parts + rx_row_update_multi +
// rx_multi_dense_update bit for bit; two launches of a few microseconds leave the step's critical path.
struct StepTail {
  const float* parts; int64_t n;
  const double* parts64; int64_t n64;
  float max_norm;
  float* scale_out; float* norm_out;
  rx_tensor_list dense;
  int dense_mode;
  float dense_lr, dense_eps;
  unsigned row_blocks;
};

template <int VEC, int CH, int MODE, bool FM>
__global__ void __launch_bounds__(RX_BLOCK)
step_tail_kernel(SegMulti m, StepTail x) {
  __shared__ double wsum[RX_BLOCK / 64];
  float tn;
  const float sc = clip_parts_block(x.parts, x.n, x.parts64, x.n64, nullptr, x.max_norm, wsum, &tn);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    x.scale_out[0] = sc;
    if (x.norm_out) x.norm_out[0] = tn;
  }
  if (blockIdx.x < x.row_blocks) {
    const int t = seg_multi_table(m);
    row_update_body<VEC, CH, MODE>(m.t[t], blockIdx.x - m.first[t], sc);
    return;
  }
  const unsigned b = blockIdx.x - x.row_blocks;
  const int k = (int)(b / RX_MULTI_BLOCKS);
  const unsigned bx = b % RX_MULTI_BLOCKS;
  if (x.dense_mode == RX_SEG_SGD) multi_dense_body<RX_SEG_SGD>(x.dense, k, bx, RX_MULTI_BLOCKS, x.dense_lr, x.dense_eps, sc);
  else multi_dense_body<RX_SEG_ADAGRAD>(x.dense, k, bx, RX_MULTI_BLOCKS, x.dense_lr, x.dense_eps, sc);
}

extern "C" int rx_step_tail(const rx_seg_table* tables, int32_t count, int mode, float lr, float eps, const rx_tensor_list* dense,
                            int dense_mode, float dense_lr, float dense_eps, const float* d_parts, int64_t n_parts,
                            const double* d_parts64, int64_t n_parts64, float max_norm, float* d_scale, float* d_total_norm,
                            int device, void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(count >= 1 && count <= SEG_MULTI_MAX && tables, "rx_step_tail: 1..%d tables per call", SEG_MULTI_MAX);
  RX_REQUIRE(mode == RX_SEG_SGD || mode == RX_SEG_ADAGRAD || mode == RX_SEG_ROWWISE_ADAGRAD, "rx_step_tail: row rule must be SGD, Adagrad or row-wise Adagrad");
  RX_REQUIRE(!dense || (dense->count >= 0 && dense->count <= RX_MULTI_MAX), "rx_step_tail: bad dense tensor list");
  RX_REQUIRE(!dense || dense->count == 0 || dense_mode == RX_SEG_SGD || dense_mode == RX_SEG_ADAGRAD, "rx_step_tail: dense rule must be SGD or Adagrad");
  RX_REQUIRE(n_parts >= 0 && (d_parts || n_parts == 0) && n_parts64 >= 0 && (d_parts64 || n_parts64 == 0) && d_scale, "rx_step_tail: bad clip arguments");
  SegMulti m{};
  RxRowShape sh0{};
  unsigned blocks = 0;
  int k = 0;
  for (int i = 0; i < count; ++i) {
    const rx_seg_table& t = tables[i];
    RX_REQUIRE(t.n >= 0 && t.dim >= 1 && t.dim <= RX_MAX_DIM, "rx_step_tail: bad shape");
    if (mode_check(mode, t.state, "rx_step_tail", false)) return 1;
    RX_REQUIRE((t.occ2 == nullptr) == (t.target2 == nullptr), "rx_step_tail: companion gradient and table come as a pair");
    RX_REQUIRE(!t.target2 || !(mode == RX_SEG_ADAGRAD || mode == RX_SEG_ROWWISE_ADAGRAD) || t.state2, "rx_step_tail: companion Adagrad state missing");
    if (t.n == 0) continue;
    RX_REQUIRE(t.occ && t.uniq_rows && t.n_uniq && t.target && t.n_rows >= 1, "rx_step_tail: null argument");
    for (int j = 0; j < k; ++j)
      RX_REQUIRE(m.t[j].target != t.target, "rx_step_tail: a table may appear once per call (two updates of one row would race)");
    RxRowShape sh;
    if (seg_shape(t.dim, t.dim, t.occ, t.target, mode == RX_SEG_ADAGRAD ? t.state : nullptr, &sh, "rx_step_tail")) return 1;
    RX_REQUIRE(k == 0 || (sh.vec == sh0.vec && sh.chunks == sh0.chunks), "rx_step_tail: the tables of one call must share the row shape");
    sh0 = sh;
    SegArgs a;
    seg_from_table(t, mode, lr, eps, nullptr, &a);
    a.occ_ld = t.dim; a.group = sh.group; a.sq_part = nullptr;
    m.t[k] = a; m.first[k] = blocks;
    blocks += (unsigned)rx_cdiv(rx_cdiv(t.n, upd_rows_of(sh.chunks, mode)) * sh.group, RX_BLOCK);
    ++k;
  }
  RX_REQUIRE(k >= 1, "rx_step_tail: every table is empty; use rx_clip_scale_parts + rx_multi_dense_update");
  m.first[k] = blocks; m.count = k;
  StepTail x{};
  x.parts = d_parts; x.n = n_parts; x.parts64 = d_parts64; x.n64 = n_parts64; x.max_norm = max_norm;
  x.scale_out = d_scale; x.norm_out = d_total_norm;
  x.dense_mode = dense_mode; x.dense_lr = dense_lr; x.dense_eps = dense_eps; x.row_blocks = blocks;
  int n_dense = 0;
  if (dense) {
    x.dense = *dense;
    n_dense = dense->count;
    for (int i = 0; i < n_dense; ++i)
      RX_REQUIRE(dense->n[i] >= 0 && ((dense->w[i] && dense->g[i] && (dense_mode == RX_SEG_SGD || dense->state[i])) || dense->n[i] == 0), "rx_step_tail: bad dense tensor %d", i);
  }
  const dim3 grid(blocks + (unsigned)n_dense * RX_MULTI_BLOCKS);
  switch (mode) {     // (the three optimizer rules only: no gradient-buffer or speculative instantiations of this kernel)
    case RX_SEG_SGD: SEG_DISPATCH_MF(RX_SEG_SGD, false, sh0, step_tail_kernel, grid, 0, (hipStream_t)stream, m, x); break;
    case RX_SEG_ADAGRAD: SEG_DISPATCH_MF(RX_SEG_ADAGRAD, false, sh0, step_tail_kernel, grid, 0, (hipStream_t)stream, m, x); break;
    default: SEG_DISPATCH_MF(RX_SEG_ROWWISE_ADAGRAD, false, sh0, step_tail_kernel, grid, 0, (hipStream_t)stream, m, x); break;
  }
  RX_LAUNCH_CHECK();
  return 0;
}

extern "C" int rx_clip_scale_parts(const float* d_parts, int64_t n_parts, const double* d_parts64, int64_t n_parts64,
                                   const double* d_acc, float max_norm, float* d_scale, float* d_total_norm,
                                   int device, void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(n_parts >= 0 && (d_parts || n_parts == 0) && n_parts64 >= 0 && (d_parts64 || n_parts64 == 0) && d_scale, "rx_clip_scale_parts: bad arguments");
  hipLaunchKernelGGL(clip_scale_parts_kernel, dim3(1), dim3(RX_BLOCK), 0, (hipStream_t)stream, d_parts, n_parts, d_parts64, n_parts64, d_acc, max_norm, d_scale, d_total_norm);
  RX_LAUNCH_CHECK();
  return 0;
}

extern "C" int rx_clip_scale(const double* d_acc, float max_norm, float* d_scale, float* d_total_norm,
                             int device, void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(d_acc && d_scale, "rx_clip_scale: null argument");
  hipLaunchKernelGGL(clip_scale_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d_acc, max_norm, d_scale, d_total_norm);
  RX_LAUNCH_CHECK();
  return 0;
}
