// Embedding row gather / tower-input assembly and its per-occurrence backward (gfx950).
//
// Rows are moved as 16-byte pieces: a group of dim/4 lanes owns one output row segment, so
// a 64-float row is one 256-byte coalesced read and a 16-float row a single 64-byte read per
// 4 lanes.  Indices are read once per group leader and broadcast with a wave shuffle.
#include "rx_common.h"

thread_local char rx_err_buf[512] = "";

extern "C" const char* rx_last_error(void) { return rx_err_buf; }
extern "C" int rx_abi_version(void) { return RX_ABI_VERSION; }

// Device wall clock (constant 100 MHz counter) written in stream order: brackets work inside a
// replayed HIP graph, where events cannot be placed between nodes after capture.
__global__ void wall_clock_kernel(int64_t* slot) { *slot = (int64_t)wall_clock64(); }

extern "C" int rx_wall_clock(int64_t* d_slot, int device, void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(d_slot, "rx_wall_clock: null slot");
  hipLaunchKernelGGL(wall_clock_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, d_slot);
  RX_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- tower-input assembly
// One launch for all segments.  A block owns GATHER_TILE consecutive output rows: it first resolves
// every index those rows need (field pick, broadcast divisor, base offset, bounds check) into an LDS
// tile with one sweep of coalesced-as-possible loads, then its threads walk the tile's 16-byte pieces
// in output order, so that consecutive lanes write consecutive pieces of one output row and read
// consecutive pieces of one table row.  A pooled segment (bag > 1) sums its bag of rows in bag order.
#define GATHER_TILE 64
#define GATHER_MAX_ENTRIES 32  // LDS index entries per output row (sum of the segments' bags)

struct GatherPlan {
  rx_gather_seg s[RX_MAX_SEGS];
  int n;
  int piece_start[RX_MAX_SEGS + 1];  // first 16-byte piece of each segment inside an output row
  int ent_start[RX_MAX_SEGS + 1];    // first index entry of each segment inside an output row
};

__global__ void __launch_bounds__(RX_BLOCK)
gather_multi_kernel(GatherPlan P, float* __restrict__ out, int64_t out_rows, int64_t out_ld, int32_t* err_flag) {
  __shared__ int64_t rows_lds[GATHER_TILE * GATHER_MAX_ENTRIES];
  __shared__ const float* tab_lds[RX_MAX_SEGS];
  __shared__ int meta_lds[RX_MAX_SEGS][4];  // dim, col, bag, first entry
  __shared__ int piece_lds[RX_MAX_SEGS + 1];
  const int64_t i0 = (int64_t)blockIdx.x * GATHER_TILE;
  const int tile = (int)(out_rows - i0 < GATHER_TILE ? out_rows - i0 : GATHER_TILE);
  const int ents = P.ent_start[P.n];
  if (threadIdx.x <= P.n) piece_lds[threadIdx.x] = P.piece_start[threadIdx.x];
  if (threadIdx.x < P.n) {
    const rx_gather_seg& sg = P.s[threadIdx.x];
    tab_lds[threadIdx.x] = sg.table;
    meta_lds[threadIdx.x][0] = sg.dim; meta_lds[threadIdx.x][1] = sg.col;
    meta_lds[threadIdx.x][2] = sg.bag > 1 ? sg.bag : 1; meta_lds[threadIdx.x][3] = P.ent_start[threadIdx.x];
  }
  // index tile: entry e of row r (r fastest, so that a wavefront reads neighbouring samples)
  for (int q = 0; q < P.n; ++q) {
    const rx_gather_seg& sg = P.s[q];
    const int bag = sg.bag > 1 ? sg.bag : 1;
    const int e0 = P.ent_start[q];
    for (int t = threadIdx.x; t < tile * bag; t += RX_BLOCK) {
      const int m = t / tile, r = t - m * tile;
      const int64_t j = ((i0 + r) / sg.div) * sg.stride + sg.k + m;
      const int64_t row = sg.idx[j] + sg.base;
      rows_lds[r * ents + e0 + m] = rx_row_ok(row, sg.rows, err_flag) ? row : -1;
    }
  }
  __syncthreads();
  const int ppr = piece_lds[P.n];
  const int total = tile * ppr;
#pragma unroll 4
  for (int p = threadIdx.x; p < total; p += RX_BLOCK) {
    const int r = p / ppr, c = p - r * ppr;
    int q = 0;
#pragma unroll
    for (int u = 1; u < RX_MAX_SEGS; ++u) q += (u < P.n && c >= piece_lds[u]) ? 1 : 0;
    const int cc = (c - piece_lds[q]) * 4;
    const int dim = meta_lds[q][0], bag = meta_lds[q][2];
    const float* tab = tab_lds[q];
    const int64_t* ent = rows_lds + r * ents + meta_lds[q][3];
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bag == 1) {  // a copy, bit for bit
      const int64_t row = ent[0];
      if (row >= 0) v = *reinterpret_cast<const float4*>(tab + row * dim + cc);
    } else {
      for (int m = 0; m < bag; ++m) {
        const int64_t row = ent[m];
        if (row >= 0) {
          const float4 w = *reinterpret_cast<const float4*>(tab + row * dim + cc);
          v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
        }
      }
    }
    *reinterpret_cast<float4*>(out + (i0 + r) * out_ld + meta_lds[q][1] + cc) = v;
  }
}

// Fallback for widths, columns or pointers that rule out 16-byte pieces: one launch per segment,
// a lane group per output row.
template <int VEC>
__global__ void __launch_bounds__(RX_BLOCK)
gather_seg_kernel(rx_gather_seg s, float* __restrict__ out, int64_t out_rows, int64_t out_ld,
                  int group, int chunks, int32_t* err_flag) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t i = gid / group;  // output row
  const int lg = (int)(gid % group);
  if (i >= out_rows) return;
  const int bag = s.bag > 1 ? s.bag : 1;
  const int64_t j = (i / s.div) * s.stride + s.k;
  float* dst = out + i * out_ld + s.col;
  for (int k = 0; k < chunks; ++k) {
    const int c = (lg + k * group) * VEC;
    if (c >= s.dim) continue;
    RxPack<VEC> acc = rx_zero<VEC>();
    for (int m = 0; m < bag; ++m) {
      // every lane of the group reads the same 8 bytes: one broadcast load from L1
      const int64_t row = s.idx[j + m] + s.base;
      if (!rx_row_ok(row, s.rows, (lg == 0 && k == 0) ? err_flag : nullptr)) continue;
      const RxPack<VEC> v = rx_load<VEC>(s.table + row * (int64_t)s.dim + c);
      if (bag == 1) { acc = v; break; }
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc.v[e] += v.v[e];
    }
    rx_store<VEC>(dst + c, acc);
  }
}

extern "C" int rx_gather_concat(const rx_gather_seg* segs, int n_segs, float* d_out,
                                int64_t out_rows, int64_t out_ld, int32_t* err_flag, int device,
                                void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(n_segs >= 1 && n_segs <= RX_MAX_SEGS, "rx_gather_concat: n_segs %d not in 1..%d", n_segs, RX_MAX_SEGS);
  RX_REQUIRE(out_rows >= 0 && out_ld >= 1, "rx_gather_concat: bad output shape");
  if (out_rows == 0) return 0;
  RX_REQUIRE(segs && d_out, "rx_gather_concat: null argument");
  hipStream_t st = (hipStream_t)stream;
  bool pieces = (out_ld % 4 == 0) && rx_aligned16(d_out);
  GatherPlan P{};
  P.n = n_segs;
  for (int q = 0; q < n_segs; ++q) {
    const rx_gather_seg& s = segs[q];
    RX_REQUIRE(s.table && s.idx, "rx_gather_concat: segment %d has a null pointer", q);
    RX_REQUIRE(s.dim >= 1 && s.dim <= RX_MAX_DIM, "rx_gather_concat: segment %d dim %d unsupported", q, s.dim);
    RX_REQUIRE(s.div >= 1 && s.stride >= 1 && s.k >= 0 && s.k < s.stride, "rx_gather_concat: segment %d bad div/stride/k", q);
    RX_REQUIRE(s.bag >= 0 && s.k + (s.bag > 1 ? s.bag : 1) <= s.stride, "rx_gather_concat: segment %d: bag %d leaves its index row (k %d, stride %d)", q, s.bag, s.k, s.stride);
    RX_REQUIRE(s.col >= 0 && (int64_t)s.col + s.dim <= out_ld, "rx_gather_concat: segment %d exceeds out_ld", q);
    P.s[q] = s;
    P.piece_start[q + 1] = P.piece_start[q] + s.dim / 4;
    P.ent_start[q + 1] = P.ent_start[q] + (s.bag > 1 ? s.bag : 1);
    pieces = pieces && s.dim % 4 == 0 && s.col % 4 == 0 && rx_aligned16(s.table);
  }
  if (pieces && P.ent_start[n_segs] <= GATHER_MAX_ENTRIES) {
    const unsigned grid = (unsigned)rx_cdiv(out_rows, GATHER_TILE);
    hipLaunchKernelGGL(gather_multi_kernel, dim3(grid), dim3(RX_BLOCK), 0, st, P, d_out, out_rows, out_ld, err_flag);
    RX_LAUNCH_CHECK();
    return 0;
  }
  for (int q = 0; q < n_segs; ++q) {
    const rx_gather_seg& s = segs[q];
    RxRowShape sh = rx_row_shape(s.dim);
    const bool vec4 = sh.vec == 4 && (s.col % 4 == 0) && (out_ld % 4 == 0) && rx_aligned16(s.table) && rx_aligned16(d_out);
    if (!vec4) {
      sh.vec = 1;
      int g = 1;
      while (g < s.dim && g < 64) g <<= 1;
      sh.group = g;
      sh.chunks = (s.dim + g - 1) / g;
    }
    const int64_t threads = out_rows * sh.group;
    const unsigned grid = (unsigned)rx_cdiv(threads, RX_BLOCK);
    if (vec4)
      hipLaunchKernelGGL(gather_seg_kernel<4>, dim3(grid), dim3(RX_BLOCK), 0, st, s, d_out, out_rows, out_ld, sh.group, sh.chunks, err_flag);
    else
      hipLaunchKernelGGL(gather_seg_kernel<1>, dim3(grid), dim3(RX_BLOCK), 0, st, s, d_out, out_rows, out_ld, sh.group, sh.chunks, err_flag);
    RX_LAUNCH_CHECK();
  }
  return 0;
}

// occ[q, :] = sum_{r<div} dout[q*div + r, col:col+dim]
__global__ void __launch_bounds__(RX_BLOCK)
gather_bwd_kernel(const float* __restrict__ dout, int64_t n_occ, int64_t out_ld, int col, int dim,
                  int div, float* __restrict__ occ) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t q = gid / dim;
  const int c = (int)(gid % dim);
  if (q >= n_occ) return;
  const float* p = dout + (q * div) * out_ld + col + c;
  float acc = 0.f;
  for (int r = 0; r < div; ++r) acc += p[(int64_t)r * out_ld];
  occ[q * dim + c] = acc;
}

extern "C" int rx_gather_concat_bwd(const float* d_dout, int64_t out_rows, int64_t out_ld,
                                    int32_t col, int32_t dim, int32_t div, float* d_occ,
                                    int device, void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(d_dout && d_occ, "rx_gather_concat_bwd: null argument");
  RX_REQUIRE(dim >= 1 && div >= 1 && col >= 0 && (int64_t)col + dim <= out_ld, "rx_gather_concat_bwd: bad shape");
  RX_REQUIRE(out_rows % div == 0, "rx_gather_concat_bwd: out_rows %lld not a multiple of div %d", (long long)out_rows, div);
  const int64_t n_occ = out_rows / div;
  if (n_occ == 0) return 0;
  const unsigned grid = (unsigned)rx_cdiv(n_occ * dim, RX_BLOCK);
  hipLaunchKernelGGL(gather_bwd_kernel, dim3(grid), dim3(RX_BLOCK), 0, (hipStream_t)stream, d_dout, n_occ, out_ld, col, dim, div, d_occ);
  RX_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- permutation of row blocks
// n_valid (may be null): device count of the leading entries of perm that are real; threads past
// it exit, so lists whose length never visits the host can be sized by their upper bound.
template <bool SCATTER>
__global__ void __launch_bounds__(RX_BLOCK)
permute_rows_kernel(const float* __restrict__ rows, const int32_t* __restrict__ perm, int64_t n,
                    const int32_t* __restrict__ n_valid, int dim, float* __restrict__ out) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t j = gid / dim;
  const int c = (int)(gid % dim);
  if (j >= n || (n_valid && j >= n_valid[0])) return;
  const int64_t p = perm[j];
  if (p < 0) {  // an entry that found no exchange slot (rx_pack_by_owner overflow): reads as a zero row, writes nothing
    if (!SCATTER) out[j * dim + c] = 0.f;
    return;
  }
  if (SCATTER)
    out[p * dim + c] = rows[j * dim + c];
  else
    out[j * dim + c] = rows[p * dim + c];
}

extern "C" int rx_unpermute_rows(const float* d_rows, const int32_t* d_perm, int64_t n,
                                 const int32_t* d_n_valid, int32_t dim, float* d_out, int device,
                                 void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(n >= 0 && dim >= 1, "rx_unpermute_rows: bad shape");
  if (n == 0) return 0;
  RX_REQUIRE(d_rows && d_perm && d_out, "rx_unpermute_rows: null argument");
  const unsigned grid = (unsigned)rx_cdiv(n * dim, RX_BLOCK);
  hipLaunchKernelGGL(permute_rows_kernel<true>, dim3(grid), dim3(RX_BLOCK), 0, (hipStream_t)stream, d_rows, d_perm, n, d_n_valid, dim, d_out);
  RX_LAUNCH_CHECK();
  return 0;
}

extern "C" int rx_permute_rows(const float* d_rows, const int32_t* d_perm, int64_t n,
                               const int32_t* d_n_valid, int32_t dim, float* d_out, int device,
                               void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(n >= 0 && dim >= 1, "rx_permute_rows: bad shape");
  if (n == 0) return 0;
  RX_REQUIRE(d_rows && d_perm && d_out, "rx_permute_rows: null argument");
  const unsigned grid = (unsigned)rx_cdiv(n * dim, RX_BLOCK);
  hipLaunchKernelGGL(permute_rows_kernel<false>, dim3(grid), dim3(RX_BLOCK), 0, (hipStream_t)stream, d_rows, d_perm, n, d_n_valid, dim, d_out);
  RX_LAUNCH_CHECK();
  return 0;
}

// Fixed-capacity send buffers: the j-th entry of the owner-grouped list goes to slot
// owner * capacity + (j - start of its owner's group); src_of[perm[j]] remembers that slot, so
// rows that come back in the same layout can be read in list order.  Group starts come from the
// device-side counts, so nothing here needs a host-side length.
// Padding slots name a valid local row each (slot i of a group -> row i % pad_rows): their
// gradient slots stay zero, so the owner's update leaves those rows bit-identical, and because the
// rows differ the owner's reduce sees many short runs instead of one giant run on a single row.
__global__ void __launch_bounds__(RX_BLOCK)
pack_fill_kernel(int64_t* __restrict__ packed, int64_t total, int64_t capacity, int64_t pad_rows) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) packed[i] = (i % capacity) % pad_rows;
}

__global__ void __launch_bounds__(RX_BLOCK)
pack_by_owner_kernel(const int64_t* __restrict__ send_local, const int32_t* __restrict__ perm,
                     const int64_t* __restrict__ counts, int G, int64_t n, int64_t capacity,
                     int64_t* __restrict__ packed, int32_t* __restrict__ src_of, int32_t* err_flag) {
  extern __shared__ int64_t starts[];  // [G + 1]
  if (threadIdx.x == 0) {
    int64_t run = 0;
    for (int g = 0; g < G; ++g) { starts[g] = run; run += counts[g]; }
    starts[G] = run;
  }
  __syncthreads();
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n || j >= starts[G]) return;
  int lo = 0, hi = G;  // owner g with starts[g] <= j < starts[g + 1]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (starts[mid] <= j) lo = mid; else hi = mid;
  }
  const int64_t off = j - starts[lo];
  if (off >= capacity) {
    // no slot for this entry: mark it, so that everything addressed through src_of skips it
    // (rx_permute_rows / rx_unpermute_rows below; the bounds checks of rx_fm2_fwd and the reduces)
    if (err_flag) atomicOr(err_flag, RX_ERR_CAPACITY);
    src_of[perm[j]] = -1;
    return;
  }
  const int64_t dest = (int64_t)lo * capacity + off;
  packed[dest] = send_local[j];
  src_of[perm[j]] = (int32_t)dest;
}

extern "C" int rx_pack_by_owner(const int64_t* d_send_local, const int32_t* d_perm, const int64_t* d_counts,
                                int32_t G, int64_t n, int64_t capacity, int64_t pad_rows, int64_t* d_packed,
                                int32_t* d_src_of, int32_t* err_flag, int device, void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(G >= 1 && G <= 4096 && n >= 0 && capacity >= 1 && pad_rows >= 1 && (int64_t)G * capacity < (1ll << 31), "rx_pack_by_owner: bad sizes");
  RX_REQUIRE(d_counts && d_packed, "rx_pack_by_owner: null argument");
  hipStream_t st = (hipStream_t)stream;
  const int64_t total = (int64_t)G * capacity;
  hipLaunchKernelGGL(pack_fill_kernel, dim3((unsigned)rx_cdiv(total, RX_BLOCK)), dim3(RX_BLOCK), 0, st, d_packed, total, capacity, pad_rows);
  RX_LAUNCH_CHECK();
  if (n == 0) return 0;
  RX_REQUIRE(d_send_local && d_perm && d_src_of, "rx_pack_by_owner: null argument");
  hipLaunchKernelGGL(pack_by_owner_kernel, dim3((unsigned)rx_cdiv(n, RX_BLOCK)), dim3(RX_BLOCK), (G + 1) * sizeof(int64_t), st, d_send_local, d_perm, d_counts, G, n, capacity, d_packed, d_src_of, err_flag);
  RX_LAUNCH_CHECK();
  return 0;
}

// Occurrence list of an FM batch in the order the FM kernels use: all (sample, field < F) entries
// first, then the last field's B entries; table row = field-local index + field offset.
__global__ void __launch_bounds__(RX_BLOCK)
fm_flat_rows_kernel(const int64_t* __restrict__ sparse, const int64_t* __restrict__ offsets, int64_t B, int fields,
                    int64_t* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * RX_BLOCK + threadIdx.x;
  if (t >= B * fields) return;
  const int64_t b = t / fields;
  const int f = (int)(t - b * fields);
  const int F = fields - 1;
  const int64_t row = sparse[t] + offsets[f];
  out[f < F ? b * F + f : B * F + b] = row;
}

extern "C" int rx_fm_flat_rows(const int64_t* d_sparse, const int64_t* d_offsets, int64_t B, int32_t fields,
                               int64_t* d_out, int device, void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(B >= 0 && fields >= 1 && fields <= 4096 && B * fields < (1ll << 40), "rx_fm_flat_rows: bad sizes");
  if (B == 0) return 0;
  RX_REQUIRE(d_sparse && d_offsets && d_out, "rx_fm_flat_rows: null argument");
  hipLaunchKernelGGL(fm_flat_rows_kernel, dim3((unsigned)rx_cdiv(B * fields, RX_BLOCK)), dim3(RX_BLOCK), 0, (hipStream_t)stream,
                     d_sparse, d_offsets, B, (int)fields, d_out);
  RX_LAUNCH_CHECK();
  return 0;
}

// occurrence -> distinct row -> exchange slot, composed on the index side: slot_of[u] = src_of[u]
// (64-bit copy of the list's slots) and occ_slot[i] = src_of[inverse[i]].
__global__ void __launch_bounds__(RX_BLOCK)
compose_slots_kernel(const int32_t* __restrict__ src_of, const int32_t* __restrict__ inverse, int64_t n_list, int64_t n_occ,
                     int64_t* __restrict__ slot_of, int64_t* __restrict__ occ_slot) {
  const int64_t t = (int64_t)blockIdx.x * RX_BLOCK + threadIdx.x;
  if (t < n_list) slot_of[t] = src_of[t];
  if (t < n_occ) {
    const int32_t u = inverse[t];
    occ_slot[t] = (u >= 0 && u < n_list) ? (int64_t)src_of[u] : -1;
  }
}

extern "C" int rx_compose_slots(const int32_t* d_src_of, const int32_t* d_inverse, int64_t n_list, int64_t n_occ,
                                int64_t* d_slot_of, int64_t* d_occ_slot, int device, void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(n_list >= 0 && n_occ >= 0 && n_list < (1ll << 31) && n_occ < (1ll << 40), "rx_compose_slots: bad sizes");
  const int64_t n = n_list > n_occ ? n_list : n_occ;
  if (n == 0) return 0;
  RX_REQUIRE(d_src_of && d_slot_of && (n_occ == 0 || (d_inverse && d_occ_slot)), "rx_compose_slots: null argument");
  hipLaunchKernelGGL(compose_slots_kernel, dim3((unsigned)rx_cdiv(n, RX_BLOCK)), dim3(RX_BLOCK), 0, (hipStream_t)stream,
                     d_src_of, d_inverse, n_list, n_occ, d_slot_of, d_occ_slot);
  RX_LAUNCH_CHECK();
  return 0;
}

// Zeroes the padding slots of a fixed-capacity exchange buffer (slot off >= counts[owner] of every
// owner group) and leaves the used ones alone: a thread per 16-byte piece of a row, the companion
// entry by the row's first thread.
__global__ void __launch_bounds__(RX_BLOCK)
zero_unused_slots_kernel(const int64_t* __restrict__ counts, int64_t total, int64_t capacity, int vec,
                         float4* __restrict__ rows, float* __restrict__ companion) {
  const int64_t t = (int64_t)blockIdx.x * RX_BLOCK + threadIdx.x;
  const int64_t slot = t / vec;
  if (slot >= total) return;
  const int piece = (int)(t - slot * vec);
  const int64_t owner = slot / capacity;
  if (slot - owner * capacity < counts[owner]) return;
  rows[slot * vec + piece] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (piece == 0 && companion) companion[slot] = 0.f;
}

extern "C" int rx_zero_unused_slots(const int64_t* d_counts, int32_t G, int64_t capacity, int32_t dim,
                                    float* d_rows, float* d_companion, int device, void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(G >= 1 && G <= 4096 && capacity >= 1 && (int64_t)G * capacity < (1ll << 31), "rx_zero_unused_slots: bad sizes");
  RX_REQUIRE(dim >= 4 && dim % 4 == 0 && dim <= 4096, "rx_zero_unused_slots: dim must be a multiple of 4 (16-byte pieces)");
  RX_REQUIRE(d_counts && d_rows, "rx_zero_unused_slots: null argument");
  RX_REQUIRE(((uintptr_t)d_rows & 15) == 0, "rx_zero_unused_slots: rows must be 16-byte aligned");
  const int64_t total = (int64_t)G * capacity;
  const int vec = dim / 4;
  hipLaunchKernelGGL(zero_unused_slots_kernel, dim3((unsigned)rx_cdiv(total * vec, RX_BLOCK)), dim3(RX_BLOCK), 0,
                     (hipStream_t)stream, d_counts, total, capacity, vec, (float4*)d_rows, d_companion);
  RX_LAUNCH_CHECK();
  return 0;
}
