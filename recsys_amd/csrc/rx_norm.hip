// BatchNorm1d over the batch dimension of the tower activations, fused with the ReLU and the
// dropout mask that follow it in the reference's towers (gfx950).
//
// x is [M, C] row-major (C = features), statistics are per column.  Every pass is a streaming read
// of whole rows: a block covers 64 columns with 16 row lanes of 16-byte loads and leaves fixed-order
// fp64 partials per slab of rows; a second, tiny kernel joins the slabs in slab order.  No atomics.
#include "rx_common.h"

#define BN_SLABS 64
#define BN_COLS 64

__device__ __forceinline__ float4 bn_ld4(const float* p, bool vec, int valid) {
  if (vec) return *reinterpret_cast<const float4*>(p);
  float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
  if (valid > 0) r.x = p[0];
  if (valid > 1) r.y = p[1];
  if (valid > 2) r.z = p[2];
  if (valid > 3) r.w = p[3];
  return r;
}

// bf16 side of the towers that keep their activations in bf16 (H16 variants below): the normalised
// output is written, and later read back as the ReLU / dropout gate, as 4 bf16 values (8 bytes);
// the gradient that leaves the backward pass is written the same way.
__device__ __forceinline__ uint32_t bn_bf16_rne(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
  u += 0x7fffu + ((u >> 16) & 1u);
  return u >> 16;
}
__device__ __forceinline__ void bn_st4_bf16(uint16_t* p, const float o[4]) {
  *reinterpret_cast<uint2*>(p) = make_uint2(bn_bf16_rne(o[0]) | (bn_bf16_rne(o[1]) << 16), bn_bf16_rne(o[2]) | (bn_bf16_rne(o[3]) << 16));
}
// the four values as floats that are non-zero exactly where the bf16 value is
__device__ __forceinline__ float4 bn_ld4_gate16(const uint16_t* p) {
  const uint2 g = *reinterpret_cast<const uint2*>(p);
  return make_float4((g.x & 0x7fffu) ? 1.f : 0.f, (g.x & 0x7fff0000u) ? 1.f : 0.f, (g.y & 0x7fffu) ? 1.f : 0.f, (g.y & 0x7fff0000u) ? 1.f : 0.f);
}

// partial[z][c][0..1] = (sum x, sum x^2) over the slab's rows
template <bool VEC>
__global__ void __launch_bounds__(RX_BLOCK)
bn_stats_kernel(const float* __restrict__ x, int64_t ldx, int64_t M, int C, int64_t rows_per_slab,
                double* __restrict__ partial) {
  __shared__ double red[16][BN_COLS + 1][2];
  const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int j0 = blockIdx.x * BN_COLS + cq * 4;
  const int valid = C - j0;
  const int64_t lo = (int64_t)blockIdx.z * rows_per_slab;
  const int64_t hi = (lo + rows_per_slab < M) ? lo + rows_per_slab : M;
  double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  if (valid > 0) {
    int64_t r = lo + rl;
    for (; r + 48 < hi; r += 64) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = bn_ld4(x + (r + 16 * u) * ldx + j0, VEC, valid);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { const double d = (double)e[i]; s[i] += d; q[i] += d * d; }
      }
    }
    for (; r < hi; r += 16) {
      const float4 v = bn_ld4(x + r * ldx + j0, VEC, valid);
      const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) { const double d = (double)e[i]; s[i] += d; q[i] += d * d; }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) { red[rl][cq * 4 + i][0] = s[i]; red[rl][cq * 4 + i][1] = q[i]; }
  __syncthreads();
  if (threadIdx.x < BN_COLS) {
    const int j = blockIdx.x * BN_COLS + threadIdx.x;
    double a = 0, b = 0;
#pragma unroll
    for (int t = 0; t < 16; ++t) { a += red[t][threadIdx.x][0]; b += red[t][threadIdx.x][1]; }
    if (j < C) {
      partial[((int64_t)blockIdx.z * C + j) * 2] = a;
      partial[((int64_t)blockIdx.z * C + j) * 2 + 1] = b;
    }
  }
}

// Joins the slab partials of 64 columns per block: four slab lanes per column (lane q adds slabs
// q, q+4, ... in order), joined as (l0 + l1) + (l2 + l3).
__device__ __forceinline__ void bn_join(const double* __restrict__ partial, int n_slabs, int C, int j, int q,
                                        double (*lds)[64][2], double* a_out, double* b_out) {
  double a = 0, b = 0;
  if (j < C)
    for (int z = q; z < n_slabs; z += 4) { a += partial[((int64_t)z * C + j) * 2]; b += partial[((int64_t)z * C + j) * 2 + 1]; }
  lds[q][threadIdx.x & 63][0] = a;
  lds[q][threadIdx.x & 63][1] = b;
  __syncthreads();
  const int e = threadIdx.x & 63;
  *a_out = (lds[0][e][0] + lds[1][e][0]) + (lds[2][e][0] + lds[3][e][0]);
  *b_out = (lds[0][e][1] + lds[1][e][1]) + (lds[2][e][1] + lds[3][e][1]);
}

// mean, 1/sqrt(var + eps) with the biased variance; running estimates with the unbiased one
__global__ void __launch_bounds__(RX_BLOCK)
bn_stats_final_kernel(const double* __restrict__ partial, int n_slabs, int64_t M, int C, float eps, float momentum,
                      float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ running_mean,
                      float* __restrict__ running_var) {
  __shared__ double lds[4][64][2];
  const int j = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
  double a, b;
  bn_join(partial, n_slabs, C, j, q, lds, &a, &b);
  if (q != 0 || j >= C) return;
  const double mu = a / (double)M;
  double var = b / (double)M - mu * mu;
  if (var < 0) var = 0;
  mean[j] = (float)mu;
  rstd[j] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) running_mean[j] = (1.f - momentum) * running_mean[j] + momentum * (float)mu;
  if (running_var) {
    const double unbiased = M > 1 ? var * ((double)M / (double)(M - 1)) : var;
    running_var[j] = (1.f - momentum) * running_var[j] + momentum * (float)unbiased;
  }
}

// out = keep * keep_scale * act(gamma * (x - mean) * rstd + beta).  Same geometry as the statistics
// pass (64 columns x 16 row lanes per block, a slab of rows per blockIdx.z): the per-column
// constants are read once per thread and no index arithmetic is left in the row loop.
template <bool VEC, bool H16 = false>
__global__ void __launch_bounds__(RX_BLOCK)
bn_apply_kernel(const float* __restrict__ x, int64_t ldx, int64_t M, int C, const float* __restrict__ mean,
                const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                int relu, const uint8_t* __restrict__ keep, float keep_scale, void* __restrict__ out_, int64_t ldo,
                int64_t rows_per_slab) {
  float* out = reinterpret_cast<float*>(out_);
  const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int j0 = blockIdx.x * BN_COLS + cq * 4;
  const int valid = C - j0;
  if (valid <= 0) return;
  const int64_t lo = (int64_t)blockIdx.z * rows_per_slab;
  const int64_t hi = (lo + rows_per_slab < M) ? lo + rows_per_slab : M;
  float sc[4], sh[4], gm[4], bt[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int j = j0 + (t < valid ? t : 0);
    sc[t] = rstd[j];
    sh[t] = mean[j];
    gm[t] = gamma ? gamma[j] : 1.f;
    bt[t] = beta ? beta[j] : 0.f;
  }
  for (int64_t r = lo + rl; r < hi; r += 32) {
    float4 v[2];
    uint32_t kp[2] = {0x01010101u, 0x01010101u};
    const bool second = r + 16 < hi;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int64_t rr = (u == 0 || second) ? r + 16 * u : r;
      v[u] = bn_ld4(x + rr * ldx + j0, VEC, valid);
      if (keep) {
        const uint8_t* kptr = keep + rr * (int64_t)C + j0;
        if (VEC) kp[u] = *reinterpret_cast<const uint32_t*>(kptr);
        else {
          kp[u] = 0;
          for (int t = 0; t < 4 && t < valid; ++t) kp[u] |= (uint32_t)kptr[t] << (8 * t);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (u == 1 && !second) continue;
      const float e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
      float o[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float y = (e[t] - sh[t]) * sc[t] * gm[t] + bt[t];
        if (relu) y = y > 0.f ? y : 0.f;
        if (keep) y = ((kp[u] >> (8 * t)) & 0xffu) ? y * keep_scale : 0.f;
        o[t] = y;
      }
      if constexpr (H16) {
        bn_st4_bf16(reinterpret_cast<uint16_t*>(out_) + (r + 16 * u) * ldo + j0, o);
        continue;
      }
      float* dst = out + (r + 16 * u) * ldo + j0;
      if (VEC) *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
      else
        for (int t = 0; t < 4 && t < valid; ++t) dst[t] = o[t];
    }
  }
}

// dz = gate ? dout * keep_scale * (out != 0) : dout ;  partial sums of (dz, dz * xhat)
template <bool VEC, bool H16 = false>
__global__ void __launch_bounds__(RX_BLOCK)
bn_bwd_sums_kernel(const float* __restrict__ dout, int64_t ldd, const float* __restrict__ x, int64_t ldx,
                   const void* __restrict__ out_, int64_t ldo, int64_t M, int C, const float* __restrict__ mean,
                   const float* __restrict__ rstd, int gate, float keep_scale, int64_t rows_per_slab,
                   double* __restrict__ partial) {
  __shared__ double red[16][BN_COLS + 1][2];
  const float* out = reinterpret_cast<const float*>(out_);
  const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int j0 = blockIdx.x * BN_COLS + cq * 4;
  const int valid = C - j0;
  const int64_t lo = (int64_t)blockIdx.z * rows_per_slab;
  const int64_t hi = (lo + rows_per_slab < M) ? lo + rows_per_slab : M;
  double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  if (valid > 0) {
    float mu[4], rs[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { const int j = j0 + (t < valid ? t : 0); mu[t] = mean[j]; rs[t] = rstd[j]; }
    for (int64_t r = lo + rl; r < hi; r += 32) {
      float4 vd[2], vx[2], vo[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int64_t rr = (r + 16 * u < hi) ? r + 16 * u : r;
        vd[u] = bn_ld4(dout + rr * ldd + j0, VEC, valid);
        vx[u] = bn_ld4(x + rr * ldx + j0, VEC, valid);
        if constexpr (H16) vo[u] = gate ? bn_ld4_gate16(reinterpret_cast<const uint16_t*>(out_) + rr * ldo + j0) : make_float4(1.f, 1.f, 1.f, 1.f);
        else vo[u] = gate ? bn_ld4(out + rr * ldo + j0, VEC, valid) : make_float4(1.f, 1.f, 1.f, 1.f);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (r + 16 * u >= hi) continue;
        const float d[4] = {vd[u].x, vd[u].y, vd[u].z, vd[u].w};
        const float xe[4] = {vx[u].x, vx[u].y, vx[u].z, vx[u].w};
        const float oe[4] = {vo[u].x, vo[u].y, vo[u].z, vo[u].w};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float dz = gate ? (oe[t] != 0.f ? d[t] * keep_scale : 0.f) : d[t];
          const float xh = (xe[t] - mu[t]) * rs[t];
          s[t] += (double)dz;
          q[t] += (double)dz * (double)xh;
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) { red[rl][cq * 4 + i][0] = s[i]; red[rl][cq * 4 + i][1] = q[i]; }
  __syncthreads();
  if (threadIdx.x < BN_COLS) {
    const int j = blockIdx.x * BN_COLS + threadIdx.x;
    double a = 0, b = 0;
#pragma unroll
    for (int t = 0; t < 16; ++t) { a += red[t][threadIdx.x][0]; b += red[t][threadIdx.x][1]; }
    if (j < C) {
      partial[((int64_t)blockIdx.z * C + j) * 2] = a;
      partial[((int64_t)blockIdx.z * C + j) * 2 + 1] = b;
    }
  }
}

// dbeta = sum dz, dgamma = sum dz xhat; coef[c] = (dbeta / M, dgamma / M) for the second pass
__global__ void __launch_bounds__(RX_BLOCK)
bn_bwd_final_kernel(const double* __restrict__ partial, int n_slabs, int64_t M, int C, float* __restrict__ dgamma,
                    float* __restrict__ dbeta, float* __restrict__ coef) {
  __shared__ double lds[4][64][2];
  const int j = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
  double a, b;
  bn_join(partial, n_slabs, C, j, q, lds, &a, &b);
  if (q != 0 || j >= C) return;
  if (dbeta) dbeta[j] = (float)a;
  if (dgamma) dgamma[j] = (float)b;
  coef[2 * j] = (float)(a / (double)M);
  coef[2 * j + 1] = (float)(b / (double)M);
}

// batch statistics: dx = gamma rstd (dz - mean(dz) - xhat mean(dz xhat)); fixed statistics: gamma rstd dz
template <bool VEC, bool H16 = false>
__global__ void __launch_bounds__(RX_BLOCK)
bn_bwd_dx_kernel(const float* __restrict__ dout, int64_t ldd, const float* __restrict__ x, int64_t ldx,
                 const void* __restrict__ out_, int64_t ldo, int64_t M, int C, const float* __restrict__ mean,
                 const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ coef,
                 int gate, float keep_scale, int batch_stats, void* __restrict__ dx_, int64_t lddx,
                 int64_t rows_per_slab) {
  const float* out = reinterpret_cast<const float*>(out_);
  float* dx = reinterpret_cast<float*>(dx_);
  const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int j0 = blockIdx.x * BN_COLS + cq * 4;
  const int valid = C - j0;
  if (valid <= 0) return;
  const int64_t lo = (int64_t)blockIdx.z * rows_per_slab;
  const int64_t hi = (lo + rows_per_slab < M) ? lo + rows_per_slab : M;
  float mu[4], rs[4], gg[4], c0[4], c1[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int j = j0 + (t < valid ? t : 0);
    mu[t] = mean[j]; rs[t] = rstd[j];
    gg[t] = (gamma ? gamma[j] : 1.f) * rstd[j];
    c0[t] = batch_stats ? coef[2 * j] : 0.f;
    c1[t] = batch_stats ? coef[2 * j + 1] : 0.f;
  }
  for (int64_t r = lo + rl; r < hi; r += 32) {
    float4 vd[2], vx[2], vo[2];
    const bool second = r + 16 < hi;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int64_t rr = (u == 0 || second) ? r + 16 * u : r;
      vd[u] = bn_ld4(dout + rr * ldd + j0, VEC, valid);
      vx[u] = bn_ld4(x + rr * ldx + j0, VEC, valid);
      if constexpr (H16) vo[u] = gate ? bn_ld4_gate16(reinterpret_cast<const uint16_t*>(out_) + rr * ldo + j0) : make_float4(1.f, 1.f, 1.f, 1.f);
      else vo[u] = gate ? bn_ld4(out + rr * ldo + j0, VEC, valid) : make_float4(1.f, 1.f, 1.f, 1.f);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (u == 1 && !second) continue;
      const float d[4] = {vd[u].x, vd[u].y, vd[u].z, vd[u].w};
      const float xe[4] = {vx[u].x, vx[u].y, vx[u].z, vx[u].w};
      const float oe[4] = {vo[u].x, vo[u].y, vo[u].z, vo[u].w};
      float o[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float dz = gate ? (oe[t] != 0.f ? d[t] * keep_scale : 0.f) : d[t];
        const float xh = (xe[t] - mu[t]) * rs[t];
        o[t] = gg[t] * (dz - c0[t] - xh * c1[t]);
      }
      if constexpr (H16) {
        bn_st4_bf16(reinterpret_cast<uint16_t*>(dx_) + (r + 16 * u) * lddx + j0, o);
        continue;
      }
      float* dst = dx + (r + 16 * u) * lddx + j0;
      if (VEC) *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
      else
        for (int t = 0; t < 4 && t < valid; ++t) dst[t] = o[t];
    }
  }
}

static inline bool bn_vec(const void* p, int64_t ld, int C) { return (ld % 4 == 0) && (C % 4 == 0) && rx_aligned16(p); }

static int bn_slabs(int64_t M, int64_t* rows_per_slab) {
  int64_t rows = rx_cdiv(M, BN_SLABS);
  rows = rx_cdiv(rows, 32) * 32;
  *rows_per_slab = rows;
  return (int)rx_cdiv(M, rows);
}

extern "C" size_t rx_bn_workspace_bytes(int64_t M, int32_t C) {
  (void)M;
  if (C < 1) return 256;
  return rx_align_up((size_t)BN_SLABS * C * 2 * sizeof(double), 256) + rx_align_up((size_t)C * 2 * sizeof(float), 256);
}

extern "C" int rx_bn_stats(const float* d_x, int64_t ldx, int64_t M, int32_t C, float eps, float momentum,
                           float* d_mean, float* d_rstd, float* d_running_mean, float* d_running_var, void* ws,
                           size_t ws_bytes, int device, void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(M >= 1 && C >= 1 && ldx >= C, "rx_bn_stats: bad shape M=%lld C=%d", (long long)M, C);
  RX_REQUIRE(d_x && d_mean && d_rstd && ws && ws_bytes >= rx_bn_workspace_bytes(M, C), "rx_bn_stats: null argument or small workspace");
  hipStream_t st = (hipStream_t)stream;
  int64_t rows;
  const int slabs = bn_slabs(M, &rows);
  const dim3 grid((unsigned)rx_cdiv(C, BN_COLS), 1, (unsigned)slabs);
  if (bn_vec(d_x, ldx, C)) hipLaunchKernelGGL(bn_stats_kernel<true>, grid, dim3(RX_BLOCK), 0, st, d_x, ldx, M, C, rows, (double*)ws);
  else hipLaunchKernelGGL(bn_stats_kernel<false>, grid, dim3(RX_BLOCK), 0, st, d_x, ldx, M, C, rows, (double*)ws);
  RX_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_stats_final_kernel, dim3((unsigned)rx_cdiv(C, 64)), dim3(RX_BLOCK), 0, st, (const double*)ws, slabs, M, C, eps, momentum, d_mean, d_rstd, d_running_mean, d_running_var);
  RX_LAUNCH_CHECK();
  return 0;
}

// streaming passes: enough slabs of rows to fill the chip at any width
static int bn_ew_slabs(int64_t M, int C, int64_t* rows_per_slab) {
  const int64_t col_blocks = rx_cdiv(C, BN_COLS);
  int64_t want = rx_cdiv(2048, col_blocks);
  int64_t rows = rx_cdiv(M, want);
  rows = rx_cdiv(rows, 32) * 32;
  *rows_per_slab = rows;
  return (int)rx_cdiv(M, rows);
}

extern "C" int rx_bn_apply(const float* d_x, int64_t ldx, int64_t M, int32_t C, const float* d_mean,
                           const float* d_rstd, const float* d_gamma, const float* d_beta, int relu,
                           const uint8_t* d_keep, float keep_scale, float* d_out, int64_t ldo, int device,
                           void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(M >= 0 && C >= 1 && ldx >= C && ldo >= C, "rx_bn_apply: bad shape");
  if (M == 0) return 0;
  RX_REQUIRE(d_x && d_mean && d_rstd && d_out, "rx_bn_apply: null argument");
  hipStream_t st = (hipStream_t)stream;
  int64_t rows;
  const int slabs = bn_ew_slabs(M, C, &rows);
  const dim3 grid((unsigned)rx_cdiv(C, BN_COLS), 1, (unsigned)slabs);
  if (bn_vec(d_x, ldx, C) && bn_vec(d_out, ldo, C))
    hipLaunchKernelGGL(bn_apply_kernel<true>, grid, dim3(RX_BLOCK), 0, st, d_x, ldx, M, C, d_mean, d_rstd, d_gamma, d_beta, relu, d_keep, keep_scale, d_out, ldo, rows);
  else
    hipLaunchKernelGGL(bn_apply_kernel<false>, grid, dim3(RX_BLOCK), 0, st, d_x, ldx, M, C, d_mean, d_rstd, d_gamma, d_beta, relu, d_keep, keep_scale, d_out, ldo, rows);
  RX_LAUNCH_CHECK();
  return 0;
}

extern "C" int rx_bn_bwd(const float* d_dout, int64_t ldd, const float* d_x, int64_t ldx, const float* d_out,
                         int64_t ldo, int64_t M, int32_t C, const float* d_mean, const float* d_rstd,
                         const float* d_gamma, int gate, float keep_scale, int batch_stats, float* d_dx,
                         int64_t lddx, float* d_dgamma, float* d_dbeta, void* ws, size_t ws_bytes, int device,
                         void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(M >= 1 && C >= 1 && ldd >= C && ldx >= C && lddx >= C && (!gate || ldo >= C), "rx_bn_bwd: bad shape");
  RX_REQUIRE(d_dout && d_x && d_mean && d_rstd && d_dx && ws && (!gate || d_out), "rx_bn_bwd: null argument");
  RX_REQUIRE(ws_bytes >= rx_bn_workspace_bytes(M, C), "rx_bn_bwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  int64_t rows;
  const int slabs = bn_slabs(M, &rows);
  double* partial = (double*)ws;
  float* coef = (float*)((char*)ws + rx_align_up((size_t)BN_SLABS * C * 2 * sizeof(double), 256));
  const bool vec = bn_vec(d_dout, ldd, C) && bn_vec(d_x, ldx, C) && bn_vec(d_dx, lddx, C) && (!gate || bn_vec(d_out, ldo, C));
  const dim3 grid((unsigned)rx_cdiv(C, BN_COLS), 1, (unsigned)slabs);
  if (vec) hipLaunchKernelGGL(bn_bwd_sums_kernel<true>, grid, dim3(RX_BLOCK), 0, st, d_dout, ldd, d_x, ldx, d_out, ldo, M, C, d_mean, d_rstd, gate, keep_scale, rows, partial);
  else hipLaunchKernelGGL(bn_bwd_sums_kernel<false>, grid, dim3(RX_BLOCK), 0, st, d_dout, ldd, d_x, ldx, d_out, ldo, M, C, d_mean, d_rstd, gate, keep_scale, rows, partial);
  RX_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_bwd_final_kernel, dim3((unsigned)rx_cdiv(C, 64)), dim3(RX_BLOCK), 0, st, (const double*)partial, slabs, M, C, d_dgamma, d_dbeta, coef);
  RX_LAUNCH_CHECK();
  int64_t erows;
  const int eslabs = bn_ew_slabs(M, C, &erows);
  const dim3 egrid((unsigned)rx_cdiv(C, BN_COLS), 1, (unsigned)eslabs);
  if (vec) hipLaunchKernelGGL(bn_bwd_dx_kernel<true>, egrid, dim3(RX_BLOCK), 0, st, d_dout, ldd, d_x, ldx, d_out, ldo, M, C, d_mean, d_rstd, d_gamma, (const float*)coef, gate, keep_scale, batch_stats, d_dx, lddx, erows);
  else hipLaunchKernelGGL(bn_bwd_dx_kernel<false>, egrid, dim3(RX_BLOCK), 0, st, d_dout, ldd, d_x, ldx, d_out, ldo, M, C, d_mean, d_rstd, d_gamma, (const float*)coef, gate, keep_scale, batch_stats, d_dx, lddx, erows);
  RX_LAUNCH_CHECK();
  return 0;
}


// ---- the same two passes for towers that keep their activations in bf16: the normalised output is
// written as bf16 only (the next matrix product rounds it to bf16 anyway; the ReLU / dropout gate
// only needs to know where it is zero), and the backward pass reads that gate and writes its
// gradient -- an operand of the two products behind it -- as bf16
static inline bool bn_vec16(const void* p, int64_t ld, int C) { return (ld % 4 == 0) && (C % 4 == 0) && ((uintptr_t)p % 8 == 0); }

extern "C" int rx_bn_apply16(const float* d_x, int64_t ldx, int64_t M, int32_t C, const float* d_mean,
                             const float* d_rstd, const float* d_gamma, const float* d_beta, int relu,
                             const uint8_t* d_keep, float keep_scale, uint16_t* d_out16, int64_t ldo, int device,
                             void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(M >= 0 && C >= 4 && C % 4 == 0 && ldx >= C && ldo >= C, "rx_bn_apply16: bad shape (C a multiple of 4)");
  if (M == 0) return 0;
  RX_REQUIRE(d_x && d_mean && d_rstd && d_out16, "rx_bn_apply16: null argument");
  RX_REQUIRE(bn_vec(d_x, ldx, C) && bn_vec16(d_out16, ldo, C), "rx_bn_apply16: rows must be 16-byte (input) and 8-byte (output) addressable");
  int64_t rows;
  const int slabs = bn_ew_slabs(M, C, &rows);
  const dim3 grid((unsigned)rx_cdiv(C, BN_COLS), 1, (unsigned)slabs);
  hipLaunchKernelGGL((bn_apply_kernel<true, true>), grid, dim3(RX_BLOCK), 0, (hipStream_t)stream, d_x, ldx, M, C, d_mean, d_rstd, d_gamma, d_beta,
                     relu, d_keep, keep_scale, (void*)d_out16, ldo, rows);
  RX_LAUNCH_CHECK();
  return 0;
}

extern "C" int rx_bn_bwd16(const float* d_dout, int64_t ldd, const float* d_x, int64_t ldx, const uint16_t* d_out16,
                           int64_t ldo, int64_t M, int32_t C, const float* d_mean, const float* d_rstd,
                           const float* d_gamma, int gate, float keep_scale, int batch_stats, uint16_t* d_dx16,
                           int64_t lddx, float* d_dgamma, float* d_dbeta, void* ws, size_t ws_bytes, int device,
                           void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(M >= 1 && C >= 4 && C % 4 == 0 && ldd >= C && ldx >= C && lddx >= C && (!gate || ldo >= C), "rx_bn_bwd16: bad shape (C a multiple of 4)");
  RX_REQUIRE(d_dout && d_x && d_mean && d_rstd && d_dx16 && ws && (!gate || d_out16), "rx_bn_bwd16: null argument");
  RX_REQUIRE(ws_bytes >= rx_bn_workspace_bytes(M, C), "rx_bn_bwd16: workspace too small");
  RX_REQUIRE(bn_vec(d_dout, ldd, C) && bn_vec(d_x, ldx, C) && bn_vec16(d_dx16, lddx, C) && (!gate || bn_vec16(d_out16, ldo, C)),
             "rx_bn_bwd16: rows must be 16-byte (fp32) and 8-byte (bf16) addressable");
  hipStream_t st = (hipStream_t)stream;
  int64_t rows;
  const int slabs = bn_slabs(M, &rows);
  double* partial = (double*)ws;
  float* coef = (float*)((char*)ws + rx_align_up((size_t)BN_SLABS * C * 2 * sizeof(double), 256));
  const dim3 grid((unsigned)rx_cdiv(C, BN_COLS), 1, (unsigned)slabs);
  hipLaunchKernelGGL((bn_bwd_sums_kernel<true, true>), grid, dim3(RX_BLOCK), 0, st, d_dout, ldd, d_x, ldx, (const void*)d_out16, ldo, M, C, d_mean, d_rstd,
                     gate, keep_scale, rows, partial);
  RX_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_bwd_final_kernel, dim3((unsigned)rx_cdiv(C, 64)), dim3(RX_BLOCK), 0, st, (const double*)partial, slabs, M, C, d_dgamma, d_dbeta, coef);
  RX_LAUNCH_CHECK();
  int64_t erows;
  const int eslabs = bn_ew_slabs(M, C, &erows);
  const dim3 egrid((unsigned)rx_cdiv(C, BN_COLS), 1, (unsigned)eslabs);
  hipLaunchKernelGGL((bn_bwd_dx_kernel<true, true>), egrid, dim3(RX_BLOCK), 0, st, d_dout, ldd, d_x, ldx, (const void*)d_out16, ldo, M, C, d_mean, d_rstd,
                     d_gamma, (const float*)coef, gate, keep_scale, batch_stats, (void*)d_dx16, lddx, erows);
  RX_LAUNCH_CHECK();
  return 0;
}
