// Fused gather + interaction kernels (gfx950): MF-family scores, GMF product, FM second
// order, BCE-with-logits.  One lane group (dim/4 lanes, 16-byte accesses) owns one sample:
// it keeps the sample's user-side row in registers, streams the N item rows through, and
// reduces with wave shuffles, so no [B,N,D] temporary ever reaches HBM.
#include "rx_common.h"

// (VEC, CH) dispatch: VEC floats per lane per step, CH steps per row.
#define RX_DISPATCH_SHAPE(sh, KERNEL, grid, block, lds, st, ...)                              \
  do {                                                                                        \
    if ((sh).vec == 4 && (sh).chunks == 1) hipLaunchKernelGGL((KERNEL<4, 1>), grid, block, lds, st, __VA_ARGS__); \
    else if ((sh).vec == 4 && (sh).chunks == 2) hipLaunchKernelGGL((KERNEL<4, 2>), grid, block, lds, st, __VA_ARGS__); \
    else if ((sh).vec == 4) hipLaunchKernelGGL((KERNEL<4, 4>), grid, block, lds, st, __VA_ARGS__); \
    else if ((sh).chunks == 1) hipLaunchKernelGGL((KERNEL<1, 1>), grid, block, lds, st, __VA_ARGS__); \
    else if ((sh).chunks == 2) hipLaunchKernelGGL((KERNEL<1, 2>), grid, block, lds, st, __VA_ARGS__); \
    else hipLaunchKernelGGL((KERNEL<1, 4>), grid, block, lds, st, __VA_ARGS__);              \
  } while (0)

static int rx_fused_shape(int dim, RxRowShape* out, const char* who) {
  RX_REQUIRE(dim >= 1 && dim <= RX_MAX_DIM, "%s: dim %d outside 1..%d", who, dim, RX_MAX_DIM);
  RxRowShape sh = rx_row_shape(dim);
  if (sh.chunks == 3) sh.chunks = 4;
  RX_REQUIRE(sh.chunks <= 4, "%s: dim %d needs a multiple of 4 above 256", who, dim);
  *out = sh;
  return 0;
}

// A row held across the lanes of a group: CH packs of VEC floats per lane.
template <int VEC, int CH>
struct RowRegs {
  RxPack<VEC> p[CH];
};

template <int VEC, int CH>
__device__ __forceinline__ RowRegs<VEC, CH> row_load(const float* base, bool ok, int lg, int group, int dim) {
  RowRegs<VEC, CH> r;
#pragma unroll
  for (int k = 0; k < CH; ++k) {
    const int c = (lg + k * group) * VEC;
    r.p[k] = (ok && c < dim) ? rx_load<VEC>(base + c) : rx_zero<VEC>();
  }
  return r;
}

template <int VEC, int CH>
__device__ __forceinline__ void row_store(float* base, const RowRegs<VEC, CH>& r, int lg, int group, int dim) {
#pragma unroll
  for (int k = 0; k < CH; ++k) {
    const int c = (lg + k * group) * VEC;
    if (c < dim) rx_store<VEC>(base + c, r.p[k]);
  }
}

template <int VEC, int CH>
__device__ __forceinline__ RowRegs<VEC, CH> row_zero() {
  RowRegs<VEC, CH> r;
#pragma unroll
  for (int k = 0; k < CH; ++k) r.p[k] = rx_zero<VEC>();
  return r;
}

#define ROW_FOREACH(k, i)              \
  _Pragma("unroll") for (int k = 0; k < CH; ++k) _Pragma("unroll") for (int i = 0; i < VEC; ++i)

// ================================================================= MF-family score
// The logits of sample b, formed by its lane group; lane 0 of the group hands each to `emit(n, logit)`.
template <int VEC, int CH, typename Emit>
__device__ __forceinline__ void mf_forward_sample(const rx_mf_args& a, int64_t b, int lg, int group, int32_t* ef, Emit emit) {
  const int dim = a.dim;
  const int64_t u = a.users[b];
  const bool u_ok = rx_row_ok(u, a.n_users, ef);
  RowRegs<VEC, CH> ue = row_load<VEC, CH>(a.user_table + u * dim, u_ok, lg, group, dim);
  if (a.occ_table) {
    const int64_t o = a.occs[b];
    const bool o_ok = rx_row_ok(o, a.n_occ, ef);
    RowRegs<VEC, CH> oe = row_load<VEC, CH>(a.occ_table + o * dim, o_ok, lg, group, dim);
    ROW_FOREACH(k, i) ue.p[k].v[i] += oe.p[k].v[i];
  }
  if (a.lin_w) {
    RowRegs<VEC, CH> w = row_load<VEC, CH>(a.lin_w, true, lg, group, dim);
    ROW_FOREACH(k, i) ue.p[k].v[i] *= w.p[k].v[i];
  }
  float base = 0.f;
  if (a.bias) base += a.bias[0];
  if (a.user_bias && u_ok) base += a.user_bias[u];
  float extra = a.lin_b ? a.lin_b[0] : 0.f;
  if (a.utemp_table) {
    const int64_t t = a.temps[b];
    const bool t_ok = rx_row_ok(t, a.n_temp, ef);
    RowRegs<VEC, CH> ut = row_load<VEC, CH>(a.utemp_table + u * dim, u_ok, lg, group, dim);
    RowRegs<VEC, CH> te = row_load<VEC, CH>(a.temp_table + t * dim, t_ok, lg, group, dim);
    float acc = 0.f;
    ROW_FOREACH(k, i) acc += ut.p[k].v[i] * te.p[k].v[i];
    extra += rx_group_sum(acc, group);
  }
  const int N = a.N;
  for (int n = 0; n < N; ++n) {
    const int64_t it = a.items[b * N + n];
    const bool ok = rx_row_ok(it, a.n_items, ef);
    RowRegs<VEC, CH> v = row_load<VEC, CH>(a.item_table + it * dim, ok, lg, group, dim);
    float acc = 0.f;
    ROW_FOREACH(k, i) acc += ue.p[k].v[i] * v.p[k].v[i];
    acc = rx_group_sum(acc, group);
    if (lg == 0) {
      float bt = base;
      if (a.item_bias && ok) bt += a.item_bias[it];
      emit(n, (bt + acc) + extra);
    }
  }
}

template <int VEC, int CH>
__global__ void __launch_bounds__(RX_BLOCK)
mf_fwd_kernel(rx_mf_args a, float* __restrict__ logits, int group, int32_t* err_flag) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t b = gid / group;
  const int lg = (int)(gid % group);
  if (b >= a.B) return;
  const int N = a.N;
  mf_forward_sample<VEC, CH>(a, b, lg, group, lg == 0 ? err_flag : nullptr, [&](int n, float x) { logits[b * N + n] = x; });
}

// Forward + loss of the MF family / GMF in one launch: the lane that holds a logit also forms its BCE term
// and dL/dlogit = (sigmoid(x) - y) / (B N) in fp64 (the arithmetic of bce_kernel); every block stores the
// fp64 sum of its terms, bce_final_kernel adds the block sums in a fixed order.
// One pair's BCE term (returned) and dL/dlogit, fp64: the arithmetic of bce_kernel.
template <bool F64>
__device__ __forceinline__ double mf_bce_pair(const void* __restrict__ labels, int64_t at, float xf, double inv_m, float* __restrict__ dlogits,
                                              float* __restrict__ logits) {
  const double x = (double)xf;
  const double y = F64 ? reinterpret_cast<const double*>(labels)[at] : (double)reinterpret_cast<const float*>(labels)[at];
  const double e = exp(-fabs(x));
  const double sig = x >= 0.0 ? 1.0 / (1.0 + e) : e / (1.0 + e);
  dlogits[at] = (float)((sig - y) * inv_m);
  if (logits) logits[at] = xf;
  return fmax(x, 0.0) - x * y + log1p(e);
}

template <int VEC, int CH, bool F64>
__global__ void __launch_bounds__(RX_BLOCK)
mf_fwd_bce_kernel(rx_mf_args a, const void* __restrict__ labels, float* __restrict__ logits, float* __restrict__ dlogits,
                  double* __restrict__ partial, int group, int staged, int32_t* err_flag) {
  __shared__ double wsum[RX_BLOCK / 64];
  extern __shared__ float mf_xs[];      // staged: [lane groups of the block][N] logits
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t b = gid / group;
  const int lg = (int)(gid % group);
  const int N = a.N;
  const double inv_m = 1.0 / ((double)a.B * (double)N);
  double term = 0.0;
  if (staged) {
    // The sample's N logits go through LDS and its lane group shares the fp64 part, lane l taking pairs l,
    // l + group, ...: left to the one lane that holds the logits, the fp64 sequence ran N times per sample
    // with a quarter of the lanes (rows of 16 floats) at work.
    float* xs = mf_xs + (threadIdx.x / group) * N;
    if (b < a.B) mf_forward_sample<VEC, CH>(a, b, lg, group, lg == 0 ? err_flag : nullptr, [&](int n, float xf) { xs[n] = xf; });
    __syncthreads();
    if (b < a.B)
      for (int n = lg; n < N; n += group) term += mf_bce_pair<F64>(labels, b * N + n, xs[n], inv_m, dlogits, logits);
  } else if (b < a.B) {
    mf_forward_sample<VEC, CH>(a, b, lg, group, lg == 0 ? err_flag : nullptr,
                               [&](int n, float xf) { term += mf_bce_pair<F64>(labels, b * N + n, xf, inv_m, dlogits, logits); });
  }
  term = rx_wave_sum_f64(term);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = term;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < RX_BLOCK / 64; ++w) t += wsum[w];
    partial[blockIdx.x] = t;
  }
}

static int mf_check(const rx_mf_args* a, const char* who) {
  RX_REQUIRE(a, "%s: null args", who);
  RX_REQUIRE(a->B >= 0 && a->N >= 1, "%s: bad batch shape B=%lld N=%d", who, (long long)a->B, a->N);
  RX_REQUIRE(a->user_table && a->item_table && a->users && a->items, "%s: tables and user/item indices are required", who);
  RX_REQUIRE(!a->occ_table || a->occs, "%s: occupation table without occupation indices", who);
  RX_REQUIRE((a->utemp_table != nullptr) == (a->temp_table != nullptr), "%s: temporal tables come as a pair", who);
  RX_REQUIRE(!a->temp_table || a->temps, "%s: temporal table without timestamp indices", who);
  RX_REQUIRE(a->n_users > 0 && a->n_items > 0, "%s: empty table", who);
  return 0;
}

extern "C" int rx_mf_score_fwd(const rx_mf_args* a, float* d_logits, int32_t* err_flag, int device,
                               void* stream) {
  RX_ENTER(device);
  if (mf_check(a, "rx_mf_score_fwd")) return 1;
  RX_REQUIRE(d_logits, "rx_mf_score_fwd: null logits");
  if (a->B == 0) return 0;
  RxRowShape sh;
  if (rx_fused_shape(a->dim, &sh, "rx_mf_score_fwd")) return 1;
  const unsigned grid = (unsigned)rx_cdiv(a->B * sh.group, RX_BLOCK);
  RX_DISPATCH_SHAPE(sh, mf_fwd_kernel, dim3(grid), dim3(RX_BLOCK), 0, (hipStream_t)stream, *a, d_logits, sh.group, err_flag);
  RX_LAUNCH_CHECK();
  return 0;
}

__global__ void bce_final_kernel(const double* __restrict__ partial, int n, int64_t M, double* __restrict__ loss);

// workspace: one fp64 partial per block
extern "C" size_t rx_mf_score_fwd_bce_workspace_bytes(int64_t B, int32_t dim) {
  (void)dim;   // a lane group is at most one wavefront: an upper bound on the grid whatever the row width
  const int64_t grid = rx_cdiv((B > 0 ? B : 1) * 64, RX_BLOCK);
  return rx_align_up((size_t)grid * sizeof(double), 256);
}

extern "C" int rx_mf_score_fwd_bce(const rx_mf_args* a, const void* d_labels, int labels_f64, float* d_logits,
                                   float* d_dlogits, double* d_loss, void* ws, size_t ws_bytes,
                                   int32_t* err_flag, int device, void* stream) {
  RX_ENTER(device);
  if (mf_check(a, "rx_mf_score_fwd_bce")) return 1;
  RX_REQUIRE(a->B >= 1, "rx_mf_score_fwd_bce: empty batch");
  RX_REQUIRE(d_labels && d_dlogits && ws, "rx_mf_score_fwd_bce: null argument");
  RX_REQUIRE(ws_bytes >= rx_mf_score_fwd_bce_workspace_bytes(a->B, a->dim) && ((uintptr_t)ws & 15) == 0, "rx_mf_score_fwd_bce: workspace too small or unaligned");
  RxRowShape sh;
  if (rx_fused_shape(a->dim, &sh, "rx_mf_score_fwd_bce")) return 1;
  const unsigned grid = (unsigned)rx_cdiv(a->B * sh.group, RX_BLOCK);
  double* partial = (double*)ws;
  hipStream_t st = (hipStream_t)stream;
  // room for every lane group's N logits in LDS (see the kernel): beyond 32 KiB the lane that holds them does the loss
  const size_t want = (size_t)(RX_BLOCK / sh.group) * (size_t)a->N * sizeof(float);
  const int staged = want <= 32 * 1024;
  const size_t xs_bytes = staged ? want : 0;
#define MF_BCE(V, C)                                                                                                                                \
  do {                                                                                                                                              \
    if (labels_f64) hipLaunchKernelGGL((mf_fwd_bce_kernel<V, C, true>), dim3(grid), dim3(RX_BLOCK), xs_bytes, st, *a, d_labels, d_logits, d_dlogits, partial, sh.group, staged, err_flag); \
    else hipLaunchKernelGGL((mf_fwd_bce_kernel<V, C, false>), dim3(grid), dim3(RX_BLOCK), xs_bytes, st, *a, d_labels, d_logits, d_dlogits, partial, sh.group, staged, err_flag);           \
  } while (0)
  if (sh.vec == 4 && sh.chunks == 1) MF_BCE(4, 1);
  else if (sh.vec == 4 && sh.chunks == 2) MF_BCE(4, 2);
  else if (sh.vec == 4) MF_BCE(4, 4);
  else if (sh.chunks == 1) MF_BCE(1, 1);
  else if (sh.chunks == 2) MF_BCE(1, 2);
  else MF_BCE(1, 4);
#undef MF_BCE
  RX_LAUNCH_CHECK();
  if (d_loss) {   // NULL: the block partials stay in ws and rx_mf_score_bwd adds them up (rx_mf_args.loss_parts)
    hipLaunchKernelGGL(bce_final_kernel, dim3(1), dim3(64), 0, st, (const double*)partial, (int)grid, a->B * (int64_t)a->N, d_loss);
    RX_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int32_t rx_mf_score_fwd_bce_partials(int64_t B, int32_t dim) {
  if (B < 1 || dim < 1 || dim > RX_MAX_DIM) return 0;
  RxRowShape sh = rx_row_shape(dim);
  return (int32_t)rx_cdiv(B * sh.group, RX_BLOCK);
}

extern "C" int64_t rx_mf_bwd_grid(int64_t B, int32_t dim) {
  if (dim < 1 || dim > RX_MAX_DIM || B <= 0) return 0;
  RxRowShape sh = rx_row_shape(dim);
  return rx_cdiv(B * sh.group, RX_BLOCK);
}

template <int VEC, int CH>
__global__ void __launch_bounds__(RX_BLOCK)
mf_bwd_kernel(rx_mf_args a, const float* __restrict__ g, float* __restrict__ user_occ,
              float* __restrict__ item_occ, float* __restrict__ utemp_occ,
              float* __restrict__ temp_occ, float* __restrict__ gsum_out,
              float* __restrict__ linw_part, int group) {
  extern __shared__ float lds_dw[];  // [groups_per_block][dim] when linw_part != NULL
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t b = gid / group;
  const int lg = (int)(gid % group);
  const int dim = a.dim;
  const bool live = b < a.B;
  const int N = a.N;
  RowRegs<VEC, CH> dw = row_zero<VEC, CH>();
  if (live) {
    const int64_t u = a.users[b];
    const bool u_ok = u >= 0 && u < a.n_users;
    RowRegs<VEC, CH> ue = row_load<VEC, CH>(a.user_table + u * dim, u_ok, lg, group, dim);
    if (a.occ_table) {
      const int64_t o = a.occs[b];
      const bool o_ok = o >= 0 && o < a.n_occ;
      RowRegs<VEC, CH> oe = row_load<VEC, CH>(a.occ_table + o * dim, o_ok, lg, group, dim);
      ROW_FOREACH(k, i) ue.p[k].v[i] += oe.p[k].v[i];
    }
    RowRegs<VEC, CH> w;
    if (a.lin_w) w = row_load<VEC, CH>(a.lin_w, true, lg, group, dim);
    RowRegs<VEC, CH> du = row_zero<VEC, CH>();
    float gs = 0.f;
    for (int n = 0; n < N; ++n) {
      const float gn = g[b * N + n];
      gs += gn;
      const int64_t it = a.items[b * N + n];
      const bool ok = it >= 0 && it < a.n_items;
      RowRegs<VEC, CH> v = row_load<VEC, CH>(a.item_table + it * dim, ok, lg, group, dim);
      RowRegs<VEC, CH> dv;
      if (a.lin_w) {
        ROW_FOREACH(k, i) {
          du.p[k].v[i] += gn * (v.p[k].v[i] * w.p[k].v[i]);
          dv.p[k].v[i] = gn * (ue.p[k].v[i] * w.p[k].v[i]);
          dw.p[k].v[i] += gn * (ue.p[k].v[i] * v.p[k].v[i]);
        }
      } else {
        ROW_FOREACH(k, i) {
          du.p[k].v[i] += gn * v.p[k].v[i];
          dv.p[k].v[i] = gn * ue.p[k].v[i];
        }
      }
      row_store<VEC, CH>(item_occ + (b * N + n) * dim, dv, lg, group, dim);
    }
    row_store<VEC, CH>(user_occ + b * dim, du, lg, group, dim);
    if (gsum_out && lg == 0) gsum_out[b] = gs;
    if (a.utemp_table) {
      const int64_t t = a.temps[b];
      const bool t_ok = t >= 0 && t < a.n_temp;
      RowRegs<VEC, CH> ut = row_load<VEC, CH>(a.utemp_table + u * dim, u_ok, lg, group, dim);
      RowRegs<VEC, CH> te = row_load<VEC, CH>(a.temp_table + t * dim, t_ok, lg, group, dim);
      ROW_FOREACH(k, i) {
        const float x = ut.p[k].v[i];
        ut.p[k].v[i] = gs * te.p[k].v[i];
        te.p[k].v[i] = gs * x;
      }
      row_store<VEC, CH>(utemp_occ + b * dim, ut, lg, group, dim);
      row_store<VEC, CH>(temp_occ + b * dim, te, lg, group, dim);
    }
  }
  if (a.loss_out && blockIdx.x == 0 && threadIdx.x < 64) {
    // the deferred end of the forward's loss: exactly bce_final_kernel (lane l takes l, l + 64, ...; fixed xor tree)
    double s = 0.0;
    for (int i = threadIdx.x; i < a.n_loss_parts; i += 64) s += a.loss_parts[i];
    s = rx_wave_sum_f64(s);
    if (threadIdx.x == 0) a.loss_out[0] = s / (double)a.loss_count;
  }
  if (linw_part) {
    // block-level sum of dw over the block's samples in group order (fixed => reproducible)
    const int gpb = blockDim.x / group;
    const int my = threadIdx.x / group;
    row_store<VEC, CH>(lds_dw + my * dim, dw, lg, group, dim);
    __syncthreads();
    for (int c = threadIdx.x; c < dim; c += blockDim.x) {
      float s = 0.f;
      for (int q = 0; q < gpb; ++q) s += lds_dw[q * dim + c];
      linw_part[(int64_t)blockIdx.x * dim + c] = s;
    }
  }
}

extern "C" int rx_mf_score_bwd(const rx_mf_args* a, const float* d_g, float* d_user_occ,
                               float* d_item_occ, float* d_utemp_occ, float* d_temp_occ,
                               float* d_gsum, float* d_linw_part, int device, void* stream) {
  RX_ENTER(device);
  if (mf_check(a, "rx_mf_score_bwd")) return 1;
  RX_REQUIRE(d_g && d_user_occ && d_item_occ, "rx_mf_score_bwd: null gradient buffer");
  RX_REQUIRE(!a->utemp_table || (d_utemp_occ && d_temp_occ), "rx_mf_score_bwd: temporal gradient buffers missing");
  RX_REQUIRE(!a->lin_w || d_linw_part, "rx_mf_score_bwd: lin_w given but no d_linw_part");
  RX_REQUIRE(!a->loss_out || (a->loss_parts && a->n_loss_parts >= 1 && a->loss_count >= 1 && a->B >= 1),
             "rx_mf_score_bwd: loss_out needs the forward's partial sums, their count and the number of logits");
  if (a->B == 0) return 0;
  RxRowShape sh;
  if (rx_fused_shape(a->dim, &sh, "rx_mf_score_bwd")) return 1;
  const unsigned grid = (unsigned)rx_cdiv(a->B * sh.group, RX_BLOCK);
  const size_t lds = a->lin_w ? (size_t)(RX_BLOCK / sh.group) * a->dim * sizeof(float) : 0;
  RX_REQUIRE(lds <= 64 * 1024, "rx_mf_score_bwd: dim %d too wide for the lin_w reduction", a->dim);
  float* part = a->lin_w ? d_linw_part : nullptr;
  RX_DISPATCH_SHAPE(sh, mf_bwd_kernel, dim3(grid), dim3(RX_BLOCK), lds, (hipStream_t)stream, *a, d_g, d_user_occ, d_item_occ, d_utemp_occ, d_temp_occ, d_gsum, part, sh.group);
  RX_LAUNCH_CHECK();
  return 0;
}

__global__ void __launch_bounds__(RX_BLOCK)
colsum_partials_kernel(const float* __restrict__ part, int64_t rows, int cols, float* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  float s = 0.f;
  for (int64_t r = 0; r < rows; ++r) s += part[r * cols + c];
  out[c] = s;
}

extern "C" int rx_colsum_partials(const float* d_part, int64_t rows, int32_t cols, float* d_out,
                                  int device, void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(d_part && d_out && rows >= 0 && cols >= 1, "rx_colsum_partials: bad arguments");
  const unsigned grid = (unsigned)rx_cdiv(cols, RX_BLOCK);
  hipLaunchKernelGGL(colsum_partials_kernel, dim3(grid), dim3(RX_BLOCK), 0, (hipStream_t)stream, d_part, rows, cols, d_out);
  RX_LAUNCH_CHECK();
  return 0;
}

// ================================================================= GMF product
template <int VEC, int CH>
__global__ void __launch_bounds__(RX_BLOCK)
gmf_fwd_kernel(const float* __restrict__ U, int64_t n_users, const float* __restrict__ V,
               int64_t n_items, const int64_t* __restrict__ users, const int64_t* __restrict__ items,
               int64_t B, int N, int dim, float* __restrict__ out, int64_t out_ld, int col, int group,
               int32_t* err_flag) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t b = gid / group;
  const int lg = (int)(gid % group);
  if (b >= B) return;
  int32_t* ef = lg == 0 ? err_flag : nullptr;
  const int64_t u = users[b];
  const bool u_ok = rx_row_ok(u, n_users, ef);
  RowRegs<VEC, CH> ue = row_load<VEC, CH>(U + u * dim, u_ok, lg, group, dim);
  for (int n = 0; n < N; ++n) {
    const int64_t it = items[b * N + n];
    const bool ok = rx_row_ok(it, n_items, ef);
    RowRegs<VEC, CH> v = row_load<VEC, CH>(V + it * dim, ok, lg, group, dim);
    ROW_FOREACH(k, i) v.p[k].v[i] *= ue.p[k].v[i];
    row_store<VEC, CH>(out + (b * N + n) * out_ld + col, v, lg, group, dim);
  }
}

template <int VEC, int CH>
__global__ void __launch_bounds__(RX_BLOCK)
gmf_bwd_kernel(const float* __restrict__ U, int64_t n_users, const float* __restrict__ V,
               int64_t n_items, const int64_t* __restrict__ users, const int64_t* __restrict__ items,
               int64_t B, int N, int dim, const float* __restrict__ dout, int64_t out_ld, int col,
               float* __restrict__ user_occ, float* __restrict__ item_occ, int group) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t b = gid / group;
  const int lg = (int)(gid % group);
  if (b >= B) return;
  const int64_t u = users[b];
  const bool u_ok = u >= 0 && u < n_users;
  RowRegs<VEC, CH> ue = row_load<VEC, CH>(U + u * dim, u_ok, lg, group, dim);
  RowRegs<VEC, CH> du = row_zero<VEC, CH>();
  for (int n = 0; n < N; ++n) {
    const int64_t it = items[b * N + n];
    const bool ok = it >= 0 && it < n_items;
    RowRegs<VEC, CH> v = row_load<VEC, CH>(V + it * dim, ok, lg, group, dim);
    RowRegs<VEC, CH> d = row_load<VEC, CH>(dout + (b * N + n) * out_ld + col, true, lg, group, dim);
    ROW_FOREACH(k, i) {
      du.p[k].v[i] += d.p[k].v[i] * v.p[k].v[i];
      d.p[k].v[i] *= ue.p[k].v[i];
    }
    row_store<VEC, CH>(item_occ + (b * N + n) * dim, d, lg, group, dim);
  }
  row_store<VEC, CH>(user_occ + b * dim, du, lg, group, dim);
}

static int gmf_shape(int dim, int64_t out_ld, int col, const void* out, RxRowShape* sh, const char* who) {
  if (rx_fused_shape(dim, sh, who)) return 1;
  if (sh->vec == 4 && !((out_ld % 4 == 0) && (col % 4 == 0) && rx_aligned16(out))) {
    // unaligned destination: fall back to scalar lanes
    sh->vec = 1;
    int g = 1;
    while (g < dim && g < 64) g <<= 1;
    sh->group = g;
    sh->chunks = (dim + g - 1) / g;
    if (sh->chunks == 3) sh->chunks = 4;
    RX_REQUIRE(sh->chunks <= 4, "%s: dim %d too wide for an unaligned destination", who, dim);
  }
  return 0;
}

extern "C" int rx_gmf_mul_fwd(const float* user_table, int64_t n_users, const float* item_table,
                              int64_t n_items, const int64_t* users, const int64_t* items, int64_t B,
                              int32_t N, int32_t dim, float* d_out, int64_t out_ld, int32_t col,
                              int32_t* err_flag, int device, void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(user_table && item_table && users && items && d_out, "rx_gmf_mul_fwd: null argument");
  RX_REQUIRE(B >= 0 && N >= 1 && col >= 0 && (int64_t)col + dim <= out_ld, "rx_gmf_mul_fwd: bad shape");
  if (B == 0) return 0;
  RxRowShape sh;
  if (gmf_shape(dim, out_ld, col, d_out, &sh, "rx_gmf_mul_fwd")) return 1;
  const unsigned grid = (unsigned)rx_cdiv(B * sh.group, RX_BLOCK);
  RX_DISPATCH_SHAPE(sh, gmf_fwd_kernel, dim3(grid), dim3(RX_BLOCK), 0, (hipStream_t)stream, user_table, n_users, item_table, n_items, users, items, B, N, dim, d_out, out_ld, col, sh.group, err_flag);
  RX_LAUNCH_CHECK();
  return 0;
}

extern "C" int rx_gmf_mul_bwd(const float* user_table, int64_t n_users, const float* item_table,
                              int64_t n_items, const int64_t* users, const int64_t* items, int64_t B,
                              int32_t N, int32_t dim, const float* d_dout, int64_t out_ld, int32_t col,
                              float* d_user_occ, float* d_item_occ, int device, void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(user_table && item_table && users && items && d_dout && d_user_occ && d_item_occ, "rx_gmf_mul_bwd: null argument");
  RX_REQUIRE(B >= 0 && N >= 1 && col >= 0 && (int64_t)col + dim <= out_ld, "rx_gmf_mul_bwd: bad shape");
  if (B == 0) return 0;
  RxRowShape sh;
  if (gmf_shape(dim, out_ld, col, d_dout, &sh, "rx_gmf_mul_bwd")) return 1;
  const unsigned grid = (unsigned)rx_cdiv(B * sh.group, RX_BLOCK);
  RX_DISPATCH_SHAPE(sh, gmf_bwd_kernel, dim3(grid), dim3(RX_BLOCK), 0, (hipStream_t)stream, user_table, n_users, item_table, n_items, users, items, B, N, dim, d_dout, out_ld, col, d_user_occ, d_item_occ, sh.group);
  RX_LAUNCH_CHECK();
  return 0;
}

// ================================================================= FM second order
// The block's index tile ([samples-per-block, F] int64, contiguous in ex_idx) is staged in
// LDS with the field offsets already applied, by one coalesced sweep of all 256 lanes.
template <int VEC, int CH>
__global__ void __launch_bounds__(RX_BLOCK)
fm2_fwd_kernel(rx_fm_args a, float* __restrict__ logits, float* __restrict__ deep_out,
               float* __restrict__ tsum, int group, int32_t* err_flag) {
  extern __shared__ int64_t lds_idx[];  // [spb][F]
  const int spb = blockDim.x / group;
  const int64_t b0 = (int64_t)blockIdx.x * spb;
  const int F = a.F, N = a.N, dim = a.dim;
  const int64_t fld = a.feat_ld ? a.feat_ld : dim, bld = a.bias_ld ? a.bias_ld : 1;   // table row strides
  {
    const int64_t rows_here = (a.B - b0 < spb) ? a.B - b0 : spb;
    const int n_here = (int)rows_here * F;
    const int64_t* src = a.ex_idx + b0 * a.ex_ld;
    for (int j = threadIdx.x; j < n_here; j += blockDim.x) {
      const int r = j / F, f = j - r * F;
      lds_idx[j] = src[r * a.ex_ld + f] + a.ex_base[f];
    }
  }
  __syncthreads();
  const int my = threadIdx.x / group;
  const int lg = threadIdx.x % group;
  const int64_t b = b0 + my;
  if (b >= a.B) return;
  int32_t* ef = lg == 0 ? err_flag : nullptr;
  const int64_t deep_ld = a.deep_ld;

  RowRegs<VEC, CH> S = row_zero<VEC, CH>(), Q = row_zero<VEC, CH>();
  float bsum = 0.f;
  for (int f = 0; f < F; ++f) {
    const int64_t row = lds_idx[my * F + f];
    const bool ok = rx_row_ok(row, a.n_feat, ef);
    RowRegs<VEC, CH> e = row_load<VEC, CH>(a.feat + row * fld, ok, lg, group, dim);
    if (a.bias && ok) bsum += a.bias[row * bld];
    ROW_FOREACH(k, i) {
      S.p[k].v[i] += e.p[k].v[i];
      Q.p[k].v[i] += e.p[k].v[i] * e.p[k].v[i];
    }
    if (deep_out)
      for (int n = 0; n < N; ++n)
        row_store<VEC, CH>(deep_out + (b * N + n) * deep_ld + (int64_t)(1 + f) * dim, e, lg, group, dim);
  }
  for (int n = 0; n < N; ++n) {
    const int64_t row = a.item_idx[b * a.item_ld + n] + a.item_base;
    const bool ok = rx_row_ok(row, a.n_feat, ef);
    RowRegs<VEC, CH> v = row_load<VEC, CH>(a.feat + row * fld, ok, lg, group, dim);
    float acc = 0.f;
    ROW_FOREACH(k, i) {
      const float s = S.p[k].v[i] + v.p[k].v[i];
      const float q = Q.p[k].v[i] + v.p[k].v[i] * v.p[k].v[i];
      acc += s * s - q;
    }
    acc = rx_group_sum(acc, group);
    if (deep_out) row_store<VEC, CH>(deep_out + (b * N + n) * deep_ld, v, lg, group, dim);
    if (tsum) {  // N == 1 (checked on the host): the sum over all of the sample's rows
      RowRegs<VEC, CH> t;
      ROW_FOREACH(k, i) t.p[k].v[i] = S.p[k].v[i] + v.p[k].v[i];
      row_store<VEC, CH>(tsum + b * dim, t, lg, group, dim);
    }
    if (lg == 0 && logits) {
      float bt = bsum;
      if (a.bias && ok) bt += a.bias[row * bld];
      logits[b * N + n] = (a.flags & RX_FM_NO_SECOND_ORDER) ? bt : bt + 0.5f * acc;
    }
  }
}

template <int VEC, int CH>
__global__ void __launch_bounds__(RX_BLOCK)
fm2_bwd_kernel(rx_fm_args a, const float* __restrict__ g, const float* __restrict__ d_deep,
               float* __restrict__ occ, float* __restrict__ bias_occ, int group) {
  extern __shared__ int64_t lds_idx[];
  const int spb = blockDim.x / group;
  const int64_t b0 = (int64_t)blockIdx.x * spb;
  const int F = a.F, N = a.N, dim = a.dim;
  const int64_t fld = a.feat_ld ? a.feat_ld : dim;   // table row stride
  {
    const int64_t rows_here = (a.B - b0 < spb) ? a.B - b0 : spb;
    const int n_here = (int)rows_here * F;
    const int64_t* src = a.ex_idx + b0 * a.ex_ld;
    for (int j = threadIdx.x; j < n_here; j += blockDim.x) {
      const int r = j / F, f = j - r * F;
      lds_idx[j] = src[r * a.ex_ld + f] + a.ex_base[f];
    }
  }
  __syncthreads();
  const int my = threadIdx.x / group;
  const int lg = threadIdx.x % group;
  const int64_t b = b0 + my;
  if (b >= a.B) return;
  const int64_t deep_ld = a.deep_ld;
  float* item_occ = occ + a.B * F * (int64_t)dim;

  // without the second-order term only the tower gradient (d_deep) reaches the rows
  const float so = (a.flags & RX_FM_NO_SECOND_ORDER) ? 0.f : 1.f;
  // pass A: S = sum_f e_f
  RowRegs<VEC, CH> S = row_zero<VEC, CH>();
  for (int f = 0; f < F; ++f) {
    const int64_t row = lds_idx[my * F + f];
    const bool ok = row >= 0 && row < a.n_feat;
    RowRegs<VEC, CH> e = row_load<VEC, CH>(a.feat + row * fld, ok, lg, group, dim);
    ROW_FOREACH(k, i) S.p[k].v[i] += e.p[k].v[i];
  }
  // pass B: items.  d v_n = g_n S (+ deep); gv = sum_n g_n v_n; G = sum_n g_n
  RowRegs<VEC, CH> gv = row_zero<VEC, CH>();
  float G = 0.f;
  for (int n = 0; n < N; ++n) {
    const float gn = g[b * N + n];
    G += gn;
    const int64_t row = a.item_idx[b * a.item_ld + n] + a.item_base;
    const bool ok = row >= 0 && row < a.n_feat;
    RowRegs<VEC, CH> v = row_load<VEC, CH>(a.feat + row * fld, ok, lg, group, dim);
    RowRegs<VEC, CH> dv;
    ROW_FOREACH(k, i) {
      gv.p[k].v[i] += (so * gn) * v.p[k].v[i];
      dv.p[k].v[i] = (so * gn) * S.p[k].v[i];
    }
    if (d_deep) {
      RowRegs<VEC, CH> dd = row_load<VEC, CH>(d_deep + (b * N + n) * deep_ld, true, lg, group, dim);
      ROW_FOREACH(k, i) dv.p[k].v[i] += dd.p[k].v[i];
    }
    row_store<VEC, CH>(item_occ + (b * N + n) * dim, dv, lg, group, dim);
    if (bias_occ && lg == 0) bias_occ[a.B * F + b * N + n] = gn;
  }
  // pass C: d e_f = G (S - e_f) + gv (+ sum_n deep)
  for (int f = 0; f < F; ++f) {
    const int64_t row = lds_idx[my * F + f];
    const bool ok = row >= 0 && row < a.n_feat;
    RowRegs<VEC, CH> e = row_load<VEC, CH>(a.feat + row * fld, ok, lg, group, dim);
    ROW_FOREACH(k, i) e.p[k].v[i] = (so * G) * (S.p[k].v[i] - e.p[k].v[i]) + gv.p[k].v[i];
    if (d_deep) {
      for (int n = 0; n < N; ++n) {
        RowRegs<VEC, CH> dd = row_load<VEC, CH>(d_deep + (b * N + n) * deep_ld + (int64_t)(1 + f) * dim, true, lg, group, dim);
        ROW_FOREACH(k, i) e.p[k].v[i] += dd.p[k].v[i];
      }
    }
    row_store<VEC, CH>(occ + (b * F + f) * dim, e, lg, group, dim);
    if (bias_occ && lg == 0) bias_occ[b * F + f] = G;
  }
}

__global__ void __launch_bounds__(RX_BLOCK)
fm2_flat_kernel(rx_fm_args a, int64_t* __restrict__ flat) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n_ex = a.B * a.F;
  if (i < n_ex)
    flat[i] = a.ex_idx[(i / a.F) * a.ex_ld + (i % a.F)] + a.ex_base[i % a.F];
  else if (i < n_ex + a.B * a.N) {
    const int64_t q = i - n_ex;
    flat[i] = a.item_idx[(q / a.N) * a.item_ld + (q % a.N)] + a.item_base;
  }
}

static int fm_check(const rx_fm_args* a, const char* who) {
  RX_REQUIRE(a, "%s: null args", who);
  RX_REQUIRE(a->B >= 0 && a->F >= 1 && a->N >= 1, "%s: bad shape B=%lld F=%d N=%d", who, (long long)a->B, a->F, a->N);
  RX_REQUIRE(a->feat && a->ex_idx && a->ex_base && a->item_idx, "%s: null table or index pointer", who);
  RX_REQUIRE(a->n_feat > 0, "%s: empty table", who);
  RX_REQUIRE(a->ex_ld >= a->F && a->item_ld >= a->N, "%s: index row strides smaller than F / N", who);
  RX_REQUIRE(a->deep_ld >= (int64_t)(1 + a->F) * a->dim, "%s: deep_ld smaller than (1+F)*dim", who);
  RX_REQUIRE(a->feat_ld == 0 || (a->feat_ld >= a->dim && (a->dim % 4 != 0 || a->feat_ld % 4 == 0)), "%s: feat_ld must be >= dim (and a multiple of 4 floats for 16-byte rows)", who);
  RX_REQUIRE(a->bias_ld >= 0, "%s: negative bias_ld", who);
  return 0;
}

static int fm_launch_shape(const rx_fm_args* a, RxRowShape* sh, size_t* lds, const char* who) {
  if (rx_fused_shape(a->dim, sh, who)) return 1;
  *lds = (size_t)(RX_BLOCK / sh->group) * a->F * sizeof(int64_t);
  RX_REQUIRE(*lds <= 96 * 1024, "%s: F=%d too many fields for the index tile", who, a->F);
  return 0;
}

extern "C" int rx_fm2_fwd(const rx_fm_args* a, float* d_logits, float* d_deep_out, float* d_tsum,
                          int32_t* err_flag, int device, void* stream) {
  RX_ENTER(device);
  if (fm_check(a, "rx_fm2_fwd")) return 1;
  RX_REQUIRE(d_logits || d_deep_out, "rx_fm2_fwd: neither logits nor deep_out requested");
  RX_REQUIRE(!d_tsum || a->N == 1, "rx_fm2_fwd: the per-sample row sum is defined for N == 1 only");
  if (a->B == 0) return 0;
  RxRowShape sh;
  size_t lds;
  if (fm_launch_shape(a, &sh, &lds, "rx_fm2_fwd")) return 1;
  RX_REQUIRE(!d_deep_out || sh.vec == 1 || (rx_aligned16(d_deep_out) && a->deep_ld % 4 == 0), "rx_fm2_fwd: deep_out must be 16-byte aligned with deep_ld a multiple of 4");
  const unsigned grid = (unsigned)rx_cdiv(a->B, RX_BLOCK / sh.group);
  RX_DISPATCH_SHAPE(sh, fm2_fwd_kernel, dim3(grid), dim3(RX_BLOCK), lds, (hipStream_t)stream, *a, d_logits, d_deep_out, d_tsum, sh.group, err_flag);
  RX_LAUNCH_CHECK();
  return 0;
}

__global__ void bce_final_kernel(const double* __restrict__ partial, int n, int64_t M, double* __restrict__ loss);

// ---- forward + loss of the one-item-per-sample FM: the lane group that has just formed a
// sample's logit also forms its BCE term and dL/dlogit g_b (fp64, as bce_kernel does), and leaves
// what the field-grouped backward reads per occurrence in one 128-byte-aligned record:
// gt[b, 0:dim] = g_b * T_b (T_b = sum of the sample's rows) and gt[b, dim] = g_b.  Every block stores
// the fp64 sum of its terms; bce_final_kernel adds the block sums in a fixed order.  (Letting the
// last block to finish do that sum behind a ticket was measured: the ticket's two memory round
// trips at the end of every block cost 12 us per launch, the separate one-block launch 4.)
template <int VEC, int CH, bool F64>
__global__ void __launch_bounds__(RX_BLOCK)
fm2_fwd_bce_kernel(rx_fm_args a, const void* __restrict__ labels, float* __restrict__ logits,
                   float* __restrict__ gt, int64_t ld_gt, double* __restrict__ partial, int group,
                   int32_t* err_flag) {
  extern __shared__ int64_t lds_idx[];  // [spb][F]
  __shared__ double wsum[RX_BLOCK / 64];
  const int spb = blockDim.x / group;
  const int64_t b0 = (int64_t)blockIdx.x * spb;
  const int F = a.F, dim = a.dim;
  const int64_t fld = a.feat_ld ? a.feat_ld : dim, bld = a.bias_ld ? a.bias_ld : 1;   // table row strides
  {
    const int64_t rows_here = (a.B - b0 < spb) ? a.B - b0 : spb;
    const int n_here = (int)rows_here * F;
    const int64_t* src = a.ex_idx + b0 * a.ex_ld;
    for (int j = threadIdx.x; j < n_here; j += blockDim.x) {
      const int r = j / F, f = j - r * F;
      lds_idx[j] = src[r * a.ex_ld + f] + a.ex_base[f];
    }
  }
  __syncthreads();
  const int my = threadIdx.x / group;
  const int lg = threadIdx.x % group;
  const int64_t b = b0 + my;
  const bool active = b < a.B;
  int32_t* ef = lg == 0 ? err_flag : nullptr;
  double term = 0.0;
  if (active) {
    RowRegs<VEC, CH> S = row_zero<VEC, CH>(), Q = row_zero<VEC, CH>();
    float bsum = 0.f;
    for (int f = 0; f < F; ++f) {
      const int64_t row = lds_idx[my * F + f];
      const bool ok = rx_row_ok(row, a.n_feat, ef);
      RowRegs<VEC, CH> e = row_load<VEC, CH>(a.feat + row * fld, ok, lg, group, dim);
      if (a.bias && ok) bsum += a.bias[row * bld];
      ROW_FOREACH(k, i) {
        S.p[k].v[i] += e.p[k].v[i];
        Q.p[k].v[i] += e.p[k].v[i] * e.p[k].v[i];
      }
    }
    const int64_t row = a.item_idx[b * a.item_ld] + a.item_base;
    const bool ok = rx_row_ok(row, a.n_feat, ef);
    RowRegs<VEC, CH> v = row_load<VEC, CH>(a.feat + row * fld, ok, lg, group, dim);
    float acc = 0.f;
    RowRegs<VEC, CH> t;
    ROW_FOREACH(k, i) {
      const float s = S.p[k].v[i] + v.p[k].v[i];
      const float q = Q.p[k].v[i] + v.p[k].v[i] * v.p[k].v[i];
      acc += s * s - q;
      t.p[k].v[i] = s;
    }
    acc = rx_group_sum(acc, group);
    float bt = bsum;
    if (a.bias && ok) bt += a.bias[row * bld];
    const float xf = (a.flags & RX_FM_NO_SECOND_ORDER) ? bt : bt + 0.5f * acc;
    float g = 0.f;
    if (lg == 0) {
      const double x = (double)xf;
      const double y = F64 ? reinterpret_cast<const double*>(labels)[b] : (double)reinterpret_cast<const float*>(labels)[b];
      const double e = exp(-fabs(x));
      term = fmax(x, 0.0) - x * y + log1p(e);
      const double sig = x >= 0.0 ? 1.0 / (1.0 + e) : e / (1.0 + e);
      g = (float)((sig - y) * (1.0 / (double)a.B));
      if (logits) logits[b] = xf;
      gt[b * ld_gt + dim] = g;
    }
    g = __shfl(g, (int)(threadIdx.x & 63) - lg, 64);
    ROW_FOREACH(k, i) t.p[k].v[i] *= g;
    row_store<VEC, CH>(gt + b * ld_gt, t, lg, group, dim);
  }
  term = rx_wave_sum_f64(term);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = term;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int w = 0; w < RX_BLOCK / 64; ++w) s += wsum[w];
    partial[blockIdx.x] = s;
  }
}

// workspace: one fp64 partial per block
extern "C" size_t rx_fm2_fwd_bce_workspace_bytes(int64_t B, int32_t dim) {
  RxRowShape sh = rx_row_shape(dim > 0 ? dim : 1);
  if (sh.chunks == 3) sh.chunks = 4;
  const int64_t grid = rx_cdiv(B > 0 ? B : 1, RX_BLOCK / sh.group);
  return rx_align_up((size_t)grid * sizeof(double), 256);
}

extern "C" int rx_fm2_fwd_bce(const rx_fm_args* a, const void* d_labels, int labels_f64, float* d_logits,
                              float* d_gt, int64_t ld_gt, double* d_loss, void* ws, size_t ws_bytes,
                              int32_t* err_flag, int device, void* stream) {
  RX_ENTER(device);
  if (fm_check(a, "rx_fm2_fwd_bce")) return 1;
  RX_REQUIRE(a->N == 1, "rx_fm2_fwd_bce: one item per sample only (N == 1)");
  RX_REQUIRE(a->B >= 1, "rx_fm2_fwd_bce: empty batch");
  RX_REQUIRE(d_labels && d_gt && ws, "rx_fm2_fwd_bce: null argument");
  RX_REQUIRE(ws_bytes >= rx_fm2_fwd_bce_workspace_bytes(a->B, a->dim) && ((uintptr_t)ws & 15) == 0, "rx_fm2_fwd_bce: workspace too small or unaligned");
  RxRowShape sh;
  size_t lds;
  if (fm_launch_shape(a, &sh, &lds, "rx_fm2_fwd_bce")) return 1;
  RX_REQUIRE(ld_gt >= a->dim + 1 && (sh.vec == 1 || (ld_gt % 4 == 0 && rx_aligned16(d_gt))), "rx_fm2_fwd_bce: gt needs a row stride >= dim + 1, a multiple of 4 floats, 16-byte aligned");
  const unsigned grid = (unsigned)rx_cdiv(a->B, RX_BLOCK / sh.group);
  double* partial = (double*)ws;
  hipStream_t st = (hipStream_t)stream;
#define FWD_BCE(V, C)                                                                                                          do {                                                                                                                           if (labels_f64) hipLaunchKernelGGL((fm2_fwd_bce_kernel<V, C, true>), dim3(grid), dim3(RX_BLOCK), lds, st, *a, d_labels, d_logits, d_gt, ld_gt, partial, sh.group, err_flag);     else hipLaunchKernelGGL((fm2_fwd_bce_kernel<V, C, false>), dim3(grid), dim3(RX_BLOCK), lds, st, *a, d_labels, d_logits, d_gt, ld_gt, partial, sh.group, err_flag);   } while (0)
  if (sh.vec == 4 && sh.chunks == 1) FWD_BCE(4, 1);
  else if (sh.vec == 4 && sh.chunks == 2) FWD_BCE(4, 2);
  else if (sh.vec == 4) FWD_BCE(4, 4);
  else if (sh.chunks == 1) FWD_BCE(1, 1);
  else if (sh.chunks == 2) FWD_BCE(1, 2);
  else FWD_BCE(1, 4);
#undef FWD_BCE
  RX_LAUNCH_CHECK();
  if (d_loss) {   // NULL: the block partials stay in ws and rx_fg_reduce adds them up (rx_fg_reduce_args.loss_parts)
    hipLaunchKernelGGL(bce_final_kernel, dim3(1), dim3(64), 0, st, (const double*)partial, (int)grid, a->B, d_loss);
    RX_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int64_t rx_fm2_fwd_bce_partials(int64_t B, int32_t dim) {
  RxRowShape sh = rx_row_shape(dim > 0 ? dim : 1);
  if (sh.chunks == 3) sh.chunks = 4;
  return rx_cdiv(B > 0 ? B : 1, RX_BLOCK / sh.group);
}

extern "C" int rx_fm2_bwd(const rx_fm_args* a, const float* d_g, const float* d_deep, float* d_occ,
                          float* d_bias_occ, int device, void* stream) {
  RX_ENTER(device);
  if (fm_check(a, "rx_fm2_bwd")) return 1;
  RX_REQUIRE(d_g && d_occ, "rx_fm2_bwd: null gradient buffer");
  if (a->B == 0) return 0;
  RxRowShape sh;
  size_t lds;
  if (fm_launch_shape(a, &sh, &lds, "rx_fm2_bwd")) return 1;
  RX_REQUIRE(!d_deep || sh.vec == 1 || (rx_aligned16(d_deep) && a->deep_ld % 4 == 0), "rx_fm2_bwd: d_deep must be 16-byte aligned with deep_ld a multiple of 4");
  const unsigned grid = (unsigned)rx_cdiv(a->B, RX_BLOCK / sh.group);
  RX_DISPATCH_SHAPE(sh, fm2_bwd_kernel, dim3(grid), dim3(RX_BLOCK), lds, (hipStream_t)stream, *a, d_g, d_deep, d_occ, d_bias_occ, sh.group);
  RX_LAUNCH_CHECK();
  return 0;
}

extern "C" int rx_fm2_flat_indices(const rx_fm_args* a, int64_t* d_flat, int device, void* stream) {
  RX_ENTER(device);
  if (fm_check(a, "rx_fm2_flat_indices")) return 1;
  RX_REQUIRE(d_flat, "rx_fm2_flat_indices: null output");
  const int64_t n = a->B * (a->F + a->N);
  if (n == 0) return 0;
  hipLaunchKernelGGL(fm2_flat_kernel, dim3((unsigned)rx_cdiv(n, RX_BLOCK)), dim3(RX_BLOCK), 0, (hipStream_t)stream, *a, d_flat);
  RX_LAUNCH_CHECK();
  return 0;
}

// ================================================================= BCE with logits
#define RX_BCE_MAX_BLOCKS 1024

template <bool F64>
__global__ void __launch_bounds__(RX_BLOCK)
bce_kernel(const float* __restrict__ logits, const void* __restrict__ labels, int64_t M,
           double* __restrict__ partial, float* __restrict__ dlogits) {
  __shared__ double wsum[RX_BLOCK / 64];
  double acc = 0.0;
  const double inv = 1.0 / (double)M;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (int64_t)gridDim.x * blockDim.x) {
    const double x = (double)logits[i];
    const double y = F64 ? reinterpret_cast<const double*>(labels)[i] : (double)reinterpret_cast<const float*>(labels)[i];
    const double e = exp(-fabs(x));
    acc += fmax(x, 0.0) - x * y + log1p(e);
    if (dlogits) {
      const double sig = x >= 0.0 ? 1.0 / (1.0 + e) : e / (1.0 + e);
      dlogits[i] = (float)((sig - y) * inv);
    }
  }
  acc = rx_wave_sum_f64(acc);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int w = 0; w < RX_BLOCK / 64; ++w) s += wsum[w];
    partial[blockIdx.x] = s;
  }
}

// one wavefront: lane l adds partial[l], partial[l+64], ... in order, then a fixed xor tree
__global__ void bce_final_kernel(const double* __restrict__ partial, int n, int64_t M, double* __restrict__ loss) {
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) s += partial[i];
  s = rx_wave_sum_f64(s);
  if (threadIdx.x == 0) loss[0] = s / (double)M;
}

extern "C" size_t rx_bce_workspace_bytes(int64_t M) {
  (void)M;
  return RX_BCE_MAX_BLOCKS * sizeof(double);
}

extern "C" int rx_bce_fwd(const float* d_logits, const void* d_labels, int labels_f64, int64_t M,
                          double* d_loss, float* d_dlogits, void* ws, size_t ws_bytes, int device,
                          void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(d_logits && d_labels && d_loss && ws, "rx_bce_fwd: null argument");
  RX_REQUIRE(M >= 1, "rx_bce_fwd: empty input");
  RX_REQUIRE(ws_bytes >= rx_bce_workspace_bytes(M), "rx_bce_fwd: workspace too small");
  const int grid = (int)std::min<int64_t>(rx_cdiv(M, RX_BLOCK), RX_BCE_MAX_BLOCKS);
  hipStream_t st = (hipStream_t)stream;
  if (labels_f64)
    hipLaunchKernelGGL(bce_kernel<true>, dim3(grid), dim3(RX_BLOCK), 0, st, d_logits, d_labels, M, (double*)ws, d_dlogits);
  else
    hipLaunchKernelGGL(bce_kernel<false>, dim3(grid), dim3(RX_BLOCK), 0, st, d_logits, d_labels, M, (double*)ws, d_dlogits);
  RX_LAUNCH_CHECK();
  hipLaunchKernelGGL(bce_final_kernel, dim3(1), dim3(64), 0, st, (const double*)ws, grid, M, d_loss);
  RX_LAUNCH_CHECK();
  return 0;
}

// ---- the same op for callers that may not allocate (HIP graph capture, tight serving loops): the
// per-block partial sums live in a file-scope device array instead of a caller workspace.  Every
// GPU holds its own copy of the array at its own address; the address is looked up once per device
// (rx_init, or the first eager call there) and only that pointer is handed to kernels (the array's name is a host-side stand-in, not a
// device address).  bce_kernel overwrites partial[0 .. grid) and bce_final_kernel reads exactly those
// entries, so nothing has to start at zero.  Calls that overlap on one GPU share the array and race.
#include <mutex>

#define RX_BCE_NOALLOC_MAX_M (1ll << 28)
#define RX_MAX_DEVICES 64

static __device__ __align__(256) double bce_scratch[RX_BCE_MAX_BLOCKS];  // 8 KiB per GPU, for the life of the process

static std::mutex bce_scratch_mutex;
static double* bce_scratch_ptr[RX_MAX_DEVICES];  // device address of bce_scratch on each GPU, filled by rx_init

// Looks the array's address up for `device`, which must be current.  The lookup may make the runtime
// load this library's code object onto the device (an allocation, and not legal while a stream of the
// thread is being captured), so it belongs to set-up: rx_init, or the first EAGER call on a device.
static int bce_scratch_resolve(int device) {
  RX_REQUIRE(device < RX_MAX_DEVICES, "rx_bce_fwd_noalloc: device ordinal %d beyond %d", device, RX_MAX_DEVICES - 1);
  std::lock_guard<std::mutex> lock(bce_scratch_mutex);
  if (!bce_scratch_ptr[device]) {
    double* p = nullptr;
    RX_HIP(hipGetSymbolAddress((void**)&p, HIP_SYMBOL(bce_scratch)));
    RX_REQUIRE(p, "rx_bce_fwd_noalloc: hipGetSymbolAddress returned a null address");
    bce_scratch_ptr[device] = p;
  }
  return 0;
}

int rx_lt_init(int device);   // rx_blaslt.hip: the device's hipBLASLt handle

extern "C" int rx_init(int device) {
  RX_ENTER(device);
  rx_lt_init(device);
  return bce_scratch_resolve(device);
}

static int bce_scratch_for(int device, hipStream_t st, double** out) {
  RX_REQUIRE(device < RX_MAX_DEVICES, "rx_bce_fwd_noalloc: device ordinal %d beyond %d", device, RX_MAX_DEVICES - 1);
  double* p;
  {
    std::lock_guard<std::mutex> lock(bce_scratch_mutex);
    p = bce_scratch_ptr[device];
  }
  if (!p) {
    // not set up on this device yet: fine for an eager call (it does the set-up now), an error under
    // capture, where the lookup's side effects are not allowed
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    RX_HIP(hipStreamIsCapturing(st, &cap));
    RX_REQUIRE(cap == hipStreamCaptureStatusNone,
               "rx_bce_fwd_noalloc: device %d is not initialised and the stream is being captured; call rx_init(%d) "
               "(or make one eager call) before recording", device, device);
    if (bce_scratch_resolve(device)) return 1;
    std::lock_guard<std::mutex> lock(bce_scratch_mutex);
    p = bce_scratch_ptr[device];
  }
  *out = p;
  return 0;
}

extern "C" int rx_bce_fwd_noalloc(const float* d_logits, const void* d_labels, int labels_f64, int64_t M,
                                  double* d_loss, float* d_dlogits, int device, void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(d_logits && d_labels && d_loss, "rx_bce_fwd_noalloc: null argument");
  RX_REQUIRE(M >= 1, "rx_bce_fwd_noalloc: empty input");
  RX_REQUIRE(M <= RX_BCE_NOALLOC_MAX_M, "rx_bce_fwd_noalloc: M=%lld beyond the supported maximum of %lld logits", (long long)M, (long long)RX_BCE_NOALLOC_MAX_M);
  double* partial = nullptr;
  hipStream_t st = (hipStream_t)stream;
  if (bce_scratch_for(device, st, &partial)) return 1;
  const int grid = (int)std::min<int64_t>(rx_cdiv(M, RX_BLOCK), RX_BCE_MAX_BLOCKS);
  if (labels_f64)
    hipLaunchKernelGGL(bce_kernel<true>, dim3(grid), dim3(RX_BLOCK), 0, st, d_logits, d_labels, M, partial, d_dlogits);
  else
    hipLaunchKernelGGL(bce_kernel<false>, dim3(grid), dim3(RX_BLOCK), 0, st, d_logits, d_labels, M, partial, d_dlogits);
  RX_LAUNCH_CHECK();
  hipLaunchKernelGGL(bce_final_kernel, dim3(1), dim3(64), 0, st, (const double*)partial, grid, M, d_loss);
  RX_LAUNCH_CHECK();
  return 0;
}
