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
};

template <int VEC, int CH>
__device__ __forceinline__ void apply_row(const SegArgs& a, Acc<VEC, CH>& g, int64_t row, int64_t u, int lg,
                                          float scale) {
  const int dim = a.dim, group = a.group;
  if (a.mode == RX_SEG_DENSE || a.mode == RX_SEG_COMPACT) {
    float* dst = a.target + (a.mode == RX_SEG_DENSE ? row : u) * (int64_t)dim;
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int c = (lg + k * group) * VEC;
      if (c < dim) rx_store<VEC>(dst + c, g.p[k]);
    }
    return;
  }
  ACC_FOREACH(k, i) g.p[k].v[i] *= scale;
  float* w = a.target + row * (int64_t)dim;
  if (a.mode == RX_SEG_SGD) {
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
  } else if (a.mode == RX_SEG_ADAGRAD) {
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
  } else {  // RX_SEG_ROWWISE_ADAGRAD
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

template <int VEC, int CH>
__global__ void __launch_bounds__(RX_BLOCK)
segreduce_kernel(SegArgs a) {
  __shared__ float sq_lds[RX_BLOCK];
  const int group = a.group, dim = a.dim;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t u = gid / group;
  const int lg = (int)(gid % group);
  const int64_t nu = a.n_uniq[0];
  const bool live = u < nu;
  float sq = 0.f;
  if (live) {
    const int64_t row = a.uniq_rows[u];
    const int lo = a.seg_off[u], hi = a.seg_off[u + 1];
    Acc<VEC, CH> g;
    ACC_FOREACH(k, i) g.p[k].v[i] = 0.f;
    for (int p = lo; p < hi; ++p) {
      const float* src = a.occ + (int64_t)a.sorted_pos[p] * a.occ_ld;
#pragma unroll
      for (int k = 0; k < CH; ++k) {
        const int c = (lg + k * group) * VEC;
        if (c < dim) {
          RxPack<VEC> x = rx_load<VEC>(src + c);
#pragma unroll
          for (int i = 0; i < VEC; ++i) g.p[k].v[i] += x.v[i];
        }
      }
    }
    if (a.sq_part) {
#pragma unroll
      for (int k = 0; k < CH; ++k) {
        const int c = (lg + k * group) * VEC;
        if (c < dim) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) sq += g.p[k].v[i] * g.p[k].v[i];
        }
      }
    }
    if (row >= 0 && row < a.n_rows) {
      const float scale = a.scale ? a.scale[0] : 1.f;
      apply_row<VEC, CH>(a, g, row, u, lg, scale);
    }
  }
  if (a.sq_part) {
    sq_lds[threadIdx.x] = sq;
    __syncthreads();
    if (threadIdx.x == 0) {
      float s = 0.f;
      for (int t = 0; t < RX_BLOCK; ++t) s += sq_lds[t];
      a.sq_part[blockIdx.x] = s;
    }
  }
}

// 16-byte accesses need 16-byte aligned rows; widths that are not a multiple of 4 use
// scalar lanes.  The launch grid is a pure function of (n, dim): rx_segreduce_grid().
static int seg_shape(int dim, int64_t occ_ld, const void* occ, const void* target, const void* state,
                     RxRowShape* sh, const char* who) {
  *sh = rx_row_shape(dim);
  if (sh->vec == 4) {
    RX_REQUIRE((occ_ld % 4 == 0) && rx_aligned16(occ) && rx_aligned16(target) && (!state || rx_aligned16(state)),
               "%s: dim %d is a multiple of 4, so buffers must be 16-byte aligned and the row stride a multiple of 4", who, dim);
  }
  if (sh->chunks == 3) sh->chunks = 4;
  RX_REQUIRE(sh->chunks <= 4, "%s: dim %d unsupported (above 256 it must be a multiple of 4)", who, dim);
  return 0;
}

extern "C" int64_t rx_segreduce_grid(int64_t n, int32_t dim) {
  if (n <= 0 || dim < 1 || dim > RX_MAX_DIM) return 0;
  return rx_cdiv(n * rx_row_shape(dim).group, RX_BLOCK);
}

#define SEG_DISPATCH(sh, KERNEL, grid, st, args)                                                   \
  do {                                                                                             \
    if ((sh).vec == 4 && (sh).chunks == 1) hipLaunchKernelGGL((KERNEL<4, 1>), grid, dim3(RX_BLOCK), 0, st, args); \
    else if ((sh).vec == 4 && (sh).chunks == 2) hipLaunchKernelGGL((KERNEL<4, 2>), grid, dim3(RX_BLOCK), 0, st, args); \
    else if ((sh).vec == 4) hipLaunchKernelGGL((KERNEL<4, 4>), grid, dim3(RX_BLOCK), 0, st, args); \
    else if ((sh).chunks == 1) hipLaunchKernelGGL((KERNEL<1, 1>), grid, dim3(RX_BLOCK), 0, st, args); \
    else if ((sh).chunks == 2) hipLaunchKernelGGL((KERNEL<1, 2>), grid, dim3(RX_BLOCK), 0, st, args); \
    else hipLaunchKernelGGL((KERNEL<1, 4>), grid, dim3(RX_BLOCK), 0, st, args);                  \
  } while (0)

static int mode_check(int mode, const float* state, const char* who, bool allow_grad_modes) {
  RX_REQUIRE(mode >= (allow_grad_modes ? RX_SEG_DENSE : RX_SEG_SGD) && mode <= RX_SEG_ROWWISE_ADAGRAD, "%s: unknown mode %d", who, mode);
  RX_REQUIRE(!(mode == RX_SEG_ADAGRAD || mode == RX_SEG_ROWWISE_ADAGRAD) || state, "%s: Adagrad needs a state buffer", who);
  return 0;
}

extern "C" int rx_segreduce(const float* d_occ, int64_t occ_ld, int64_t n, int32_t dim,
                            const int32_t* d_sorted_pos, const int64_t* d_uniq_rows,
                            const int32_t* d_seg_off, const int32_t* d_n_uniq, int mode,
                            float* d_target, float* d_state, int64_t n_rows, float lr, float eps,
                            const float* d_scale, float* d_sq_part, int device, void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(n >= 0 && dim >= 1 && dim <= RX_MAX_DIM && occ_ld >= dim, "rx_segreduce: bad shape n=%lld dim=%d ld=%lld", (long long)n, dim, (long long)occ_ld);
  if (mode_check(mode, d_state, "rx_segreduce", true)) return 1;
  if (n == 0) return 0;
  RX_REQUIRE(d_occ && d_sorted_pos && d_uniq_rows && d_seg_off && d_n_uniq && d_target, "rx_segreduce: null argument");
  RX_REQUIRE(n_rows >= 1, "rx_segreduce: n_rows must be positive");
  RxRowShape sh;
  if (seg_shape(dim, occ_ld, d_occ, d_target, mode == RX_SEG_ADAGRAD ? d_state : nullptr, &sh, "rx_segreduce")) return 1;
  SegArgs a{d_occ, occ_ld, n, dim, d_sorted_pos, d_uniq_rows, d_seg_off, d_n_uniq, mode, d_target, d_state, n_rows, lr, eps, d_scale, d_sq_part, sh.group};
  // the grid always matches rx_segreduce_grid() so that d_sq_part has one slot per block
  const unsigned grid = (unsigned)rx_segreduce_grid(n, dim);
  SEG_DISPATCH(sh, segreduce_kernel, dim3(grid), (hipStream_t)stream, a);
  RX_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- update from a compact gradient
template <int VEC, int CH>
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
  apply_row<VEC, CH>(a, g, row, u, lg, a.scale ? a.scale[0] : 1.f);
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
  SegArgs a{d_compact, dim, n_max, dim, nullptr, d_uniq_rows, nullptr, d_n_uniq, mode, d_w, d_state, n_rows, lr, eps, d_scale, nullptr, sh.group};
  const unsigned grid = (unsigned)rx_cdiv(n_max * sh.group, RX_BLOCK);
  SEG_DISPATCH(sh, row_update_kernel, dim3(grid), (hipStream_t)stream, a);
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
__global__ void reduce_final_kernel(const double* __restrict__ partial, int n, double* __restrict__ acc,
                                    float* __restrict__ out_f32) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double s = 0.0;
  for (int i = 0; i < n; ++i) s += partial[i];
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
