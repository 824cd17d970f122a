// Dense layers of the towers on the matrix cores (gfx950): Linear forward with fused
// bias + ReLU, and both Linear gradients.  Exact-fp32 path on v_mfma_f32_32x32x2_f32 (a
// k-ordered fmaf chain, so results track an fp32 CPU GEMM to rounding) and a bf16-operand
// path on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.
//
// One kernel template covers the three products through element strides:
//   forward  Y  = X  W^T   A(i,k)=X[i,k]     B(k,j)=W[j,k]    both K-contiguous
//   grad X   dX = dY W     A(i,k)=dY[i,k]    B(k,j)=W[k,j]    A K-contiguous, B N-contiguous
//   grad W   dW = dY^T X   A(i,k)=dY[k,i]    B(k,j)=X[k,j]    both M/N-contiguous, K = batch
// The batch-long reduction of grad W is cut into slabs (one per blockIdx.z) that a second
// kernel adds in slab order: fixed order, no float atomics.
//
// LDS tiles are k-major ([BK][BM+4]): the MFMA operand read (lane -> row lane&31 of k-slice
// lane>>5) is then 32 consecutive floats per half-wave, conflict free, and the +4 pad keeps
// 16-byte alignment for vector stores while spreading transposing stores over the banks.
#include "rx_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

#define GEMM_BK 16
#define GEMM_PAD 4

struct GemmArgs {
  const float* A; int64_t a_rs, a_cs;  // A(i,k) = A[i*a_rs + k*a_cs]
  const float* B; int64_t b_rs, b_cs;  // B(k,j) = B[k*b_rs + j*b_cs]
  float* C; int64_t c_rs;              // C(i,j) = C[i*c_rs + j]
  const float* bias;                   // [N] or null
  int64_t M; int64_t N; int64_t K;
  int act;                             // 1 = ReLU
  int accumulate;                      // C += instead of C =
  int64_t k_per_slab;                  // K range per blockIdx.z
  int64_t c_slab_stride;               // C offset per blockIdx.z
  int a_vec, b_vec;                    // 16-byte global loads are legal for this operand
  int tiles_n;
  float* colsum;                       // deep-step kernels with an M-contiguous A only: [slab][M] sums of A over the slab's k
                                       // range (the bias gradient of a weight-gradient product, from the dY tiles it stages)
};

__device__ __forceinline__ uint16_t f32_to_bf16_rne(float f) {
  // round to nearest even; a NaN stays a (quiet) NaN
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t b) { return __uint_as_float(b << 16); }
__device__ __forceinline__ float round_bf16(float v) { return bf16_bits_to_f32(f32_to_bf16_rne(v)); }

// Row of flat piece index i at d pieces per row.  Streaming kernels split an index per piece; the 64-bit
// division costs more than the piece itself, so totals below 2^31 (`small`, uniform) divide in 32 bits.
__device__ __forceinline__ int64_t rx_row_of(int64_t i, int d, bool small) {
  return small ? (int64_t)((uint32_t)i / (uint32_t)d) : i / d;
}

// Load 4 consecutive elements (along the contiguous axis) with bounds guards.
__device__ __forceinline__ void load4(const float* p, int64_t stride_unused, int64_t first, int64_t limit,
                                      bool row_ok, bool vec, float out[4]) {
  (void)stride_unused;
  if (row_ok && vec && first + 3 < limit) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    out[0] = t.x; out[1] = t.y; out[2] = t.z; out[3] = t.w;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) out[e] = (row_ok && first + e < limit) ? p[e] : 0.f;
  }
}

// BK: k extent staged per iteration.  The loop is single-buffered (load, barrier, multiply, barrier), so
// every iteration costs a full trip to memory; skinny products (K <= 128: the ML-1M towers, and every
// 128-row weight-gradient slab) take BK = 64, i.e. one or two trips instead of four or eight, with all
// of a trip's loads in flight together.  Products are accumulated in ascending k whatever BK is: same bits.
template <bool A_KC, bool B_KC, int BM, int BN, int WAVES_M, int WAVES_N, int BK = GEMM_BK>
__global__ void __launch_bounds__(RX_BLOCK)
gemm_f32_kernel(GemmArgs g) {
  constexpr int TM = BM / (32 * WAVES_M);
  constexpr int TN = BN / (32 * WAVES_N);
  // k-major LDS tiles.  A K-contiguous operand is transposed on its way in (four scalar stores per 16-byte
  // piece): with 16 threads along k (deep steps) a row pitch of 1 mod 16 floats spreads a wavefront's
  // stores over all 64 banks; tiles stored as whole 16-byte pieces need a pitch that is a multiple of 4
  constexpr int LDA = BM + ((A_KC && BK > GEMM_BK) ? 1 : GEMM_PAD);
  constexpr int LDB = BN + ((B_KC && BK > GEMM_BK) ? 1 : GEMM_PAD);
  __shared__ __attribute__((aligned(16))) float As[BK * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[BK * LDB];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wave_m = wave / WAVES_N;
  const int wave_n = wave % WAVES_N;
  const int64_t tile_m = blockIdx.x / g.tiles_n;
  const int tile_n = blockIdx.x % g.tiles_n;
  const int64_t m0 = tile_m * BM;
  const int64_t n0 = (int64_t)tile_n * BN;
  const int64_t k_begin = (int64_t)blockIdx.z * g.k_per_slab;
  const int64_t k_end = (k_begin + g.k_per_slab < g.K) ? k_begin + g.k_per_slab : g.K;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);   // this thread's share of the column sums of A (deep steps, M-contiguous A)
  for (int64_t k0 = k_begin; k0 < k_end; k0 += BK) {
    if constexpr (BK > GEMM_BK) {
      // Deep steps (aligned operands only, checked by the launcher): every 16-byte piece of both tiles is
      // loaded first -- from a clamped address, without a branch, so that all loads of the trip are in
      // flight together -- and only then zeroed where it lies outside the matrix and written to LDS.
      // (The guarded loads below sit in branches of their own and therefore wait for one another.)
      constexpr int A_TPR = A_KC ? BK / 4 : BM / 4, A_PP = RX_BLOCK / A_TPR, A_NP = (A_KC ? BM : BK) / A_PP;
      constexpr int B_TPR = B_KC ? BK / 4 : BN / 4, B_PP = RX_BLOCK / B_TPR, B_NP = (B_KC ? BN : BK) / B_PP;
      static_assert(A_NP >= 1 && B_NP >= 1 && A_NP <= 16 && B_NP <= 16, "tile does not divide over the block");
      float4 va[A_NP], vb[B_NP];
      unsigned oka = 0u, okb = 0u;
      const int aq = (tid % A_TPR) * 4, bq = (tid % B_TPR) * 4;
      (void)aq; (void)bq;
#pragma unroll
      for (int p = 0; p < A_NP; ++p) {
        const int o = tid / A_TPR + A_PP * p;                 // row of the tile (A_KC) or k-row
        const int64_t row = A_KC ? m0 + o : m0 + aq, k = A_KC ? k0 + aq : k0 + o;
        const bool ok = (A_KC ? (row < g.M && k + 3 < k_end) : (k < k_end && row + 3 < g.M));
        const float* src = A_KC ? g.A + (ok ? row : 0) * g.a_rs + (ok ? k : 0) : g.A + (ok ? k : 0) * g.a_cs + (ok ? row : 0);
        va[p] = *reinterpret_cast<const float4*>(src);
        oka |= (ok ? 1u : 0u) << p;
      }
#pragma unroll
      for (int p = 0; p < B_NP; ++p) {
        const int o = tid / B_TPR + B_PP * p;
        const int64_t col = B_KC ? n0 + o : n0 + bq, k = B_KC ? k0 + bq : k0 + o;
        const bool ok = (B_KC ? (col < g.N && k + 3 < k_end) : (k < k_end && col + 3 < g.N));
        const float* src = B_KC ? g.B + (ok ? col : 0) * g.b_cs + (ok ? k : 0) : g.B + (ok ? k : 0) * g.b_rs + (ok ? col : 0);
        vb[p] = *reinterpret_cast<const float4*>(src);
        okb |= (ok ? 1u : 0u) << p;
      }
#pragma unroll
      for (int p = 0; p < A_NP; ++p) {
        const int o = tid / A_TPR + A_PP * p;
        const float4 v = ((oka >> p) & 1u) ? va[p] : make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (!A_KC) { csum.x += v.x; csum.y += v.y; csum.z += v.z; csum.w += v.w; }   // k ascending per thread
        if constexpr (A_KC) { As[(aq + 0) * LDA + o] = v.x; As[(aq + 1) * LDA + o] = v.y; As[(aq + 2) * LDA + o] = v.z; As[(aq + 3) * LDA + o] = v.w; }
        else *reinterpret_cast<float4*>(&As[o * LDA + aq]) = v;
      }
#pragma unroll
      for (int p = 0; p < B_NP; ++p) {
        const int o = tid / B_TPR + B_PP * p;
        const float4 v = ((okb >> p) & 1u) ? vb[p] : make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (B_KC) { Bs[(bq + 0) * LDB + o] = v.x; Bs[(bq + 1) * LDB + o] = v.y; Bs[(bq + 2) * LDB + o] = v.z; Bs[(bq + 3) * LDB + o] = v.w; }
        else *reinterpret_cast<float4*>(&Bs[o * LDB + bq]) = v;
      }
    } else {
    // ---- stage A tile [BM x BK] -> As[k][i]
    if (A_KC) {
      constexpr int TPR = BK / 4;          // threads per row
      constexpr int RPP = RX_BLOCK / TPR;       // rows per pass
      const int kq = (tid % TPR) * 4;
#pragma unroll
      for (int r = tid / TPR; r < BM; r += RPP) {
        const int64_t row = m0 + r;
        float v[4];
        load4(g.A + row * g.a_rs + (k0 + kq), 0, k0 + kq, k_end, row < g.M, g.a_vec, v);
#pragma unroll
        for (int e = 0; e < 4; ++e) As[(kq + e) * LDA + r] = v[e];
      }
    } else {
      constexpr int TPR = BM / 4;               // threads per k-row
      constexpr int KPP = RX_BLOCK / TPR;       // k-rows per pass
      const int iq = (tid % TPR) * 4;
#pragma unroll
      for (int kk = tid / TPR; kk < BK; kk += KPP) {
        const int64_t k = k0 + kk;
        float v[4];
        load4(g.A + k * g.a_cs + (m0 + iq), 0, m0 + iq, g.M, k < k_end, g.a_vec, v);
        *reinterpret_cast<float4*>(&As[kk * LDA + iq]) = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
    // ---- stage B tile [BK x BN] -> Bs[k][j]
    if (B_KC) {
      constexpr int TPR = BK / 4;
      constexpr int RPP = RX_BLOCK / TPR;
      const int kq = (tid % TPR) * 4;
#pragma unroll
      for (int r = tid / TPR; r < BN; r += RPP) {
        const int64_t col = n0 + r;
        float v[4];
        load4(g.B + col * g.b_cs + (k0 + kq), 0, k0 + kq, k_end, col < g.N, g.b_vec, v);
#pragma unroll
        for (int e = 0; e < 4; ++e) Bs[(kq + e) * LDB + r] = v[e];
      }
    } else {
      constexpr int TPR = BN / 4;
      constexpr int KPP = RX_BLOCK / TPR;
      const int jq = (tid % TPR) * 4;
#pragma unroll
      for (int kk = tid / TPR; kk < BK; kk += KPP) {
        const int64_t k = k0 + kk;
        float v[4];
        load4(g.B + k * g.b_rs + (n0 + jq), 0, n0 + jq, g.N, k < k_end, g.b_vec, v);
        *reinterpret_cast<float4*>(&Bs[kk * LDB + jq]) = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
    }
    __syncthreads();
    // ---- BK / 2 k-pairs of 32x32x2 MFMAs
    const int h = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = As[(kk + h) * LDA + (wave_m * TM + i) * 32 + l31];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = Bs[(kk + h) * LDB + (wave_n * TN + j) * 32 + l31];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  if constexpr (BK > GEMM_BK && !A_KC) {
    // column sums of A over this slab (block-uniform condition; the K loop ended with a barrier, so the A
    // tile's LDS is free): the threads that staged one 16-byte column piece add up in thread order
    if (g.colsum && tile_n == 0) {
      constexpr int A_TPR = BM / 4, A_PP = RX_BLOCK / A_TPR;
      const int aq = (tid % A_TPR) * 4;
      *reinterpret_cast<float4*>(&As[(tid / A_TPR) * BM + aq]) = csum;
      __syncthreads();
      if (tid < BM && m0 + tid < g.M) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < A_PP; ++q) t += As[q * BM + tid];
        g.colsum[(int64_t)blockIdx.z * g.M + m0 + tid] = t;
      }
    }
  }

  // ---- epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  float* Cb = g.C + (int64_t)blockIdx.z * g.c_slab_stride;
  const int h = lane >> 5, l31 = lane & 31;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int64_t col = n0 + (wave_n * TN + j) * 32 + l31;
    if (col >= g.N) continue;
    const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + (wave_m * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < g.M) {
          float v = acc[i][j][r] + bv;
          if (g.act == 1) v = v > 0.f ? v : 0.f;
          float* dst = Cb + row * g.c_rs + col;
          *dst = g.accumulate ? (*dst + v) : v;
        }
      }
    }
  }
}

// bf16-operand variant: operands are rounded to bf16 while staging; LDS tiles are row-major
// [BM][BK + 8] bf16 so each lane's 8 consecutive k values are one 16-byte read.
#define GEMM_BK16 32
template <bool A_KC, bool B_KC, int BM, int BN, int WAVES_M, int WAVES_N>
__global__ void __launch_bounds__(RX_BLOCK)
gemm_bf16_kernel(GemmArgs g) {
  constexpr int TM = BM / (32 * WAVES_M);
  constexpr int TN = BN / (32 * WAVES_N);
  constexpr int LDK = GEMM_BK16 + 8;  // bf16 elements per LDS row (80 bytes: 16-byte aligned, odd multiple of 16)
  __shared__ __attribute__((aligned(16))) uint16_t As[BM * LDK];
  __shared__ __attribute__((aligned(16))) uint16_t Bs[BN * LDK];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wave_m = wave / WAVES_N;
  const int wave_n = wave % WAVES_N;
  const int64_t tile_m = blockIdx.x / g.tiles_n;
  const int tile_n = blockIdx.x % g.tiles_n;
  const int64_t m0 = tile_m * BM;
  const int64_t n0 = (int64_t)tile_n * BN;
  const int64_t k_begin = (int64_t)blockIdx.z * g.k_per_slab;
  const int64_t k_end = (k_begin + g.k_per_slab < g.K) ? k_begin + g.k_per_slab : g.K;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  for (int64_t k0 = k_begin; k0 < k_end; k0 += GEMM_BK16) {
    if (A_KC) {
      constexpr int TPR = GEMM_BK16 / 4;
      constexpr int RPP = RX_BLOCK / TPR;
      const int kq = (tid % TPR) * 4;
#pragma unroll
      for (int r = tid / TPR; r < BM; r += RPP) {
        const int64_t row = m0 + r;
        float v[4];
        load4(g.A + row * g.a_rs + (k0 + kq), 0, k0 + kq, k_end, row < g.M, g.a_vec, v);
        ushort4 o = make_ushort4(f32_to_bf16_rne(v[0]), f32_to_bf16_rne(v[1]), f32_to_bf16_rne(v[2]), f32_to_bf16_rne(v[3]));
        *reinterpret_cast<ushort4*>(&As[r * LDK + kq]) = o;
      }
    } else {
      constexpr int TPR = BM / 4;
      constexpr int KPP = RX_BLOCK / TPR;
      const int iq = (tid % TPR) * 4;
#pragma unroll
      for (int kk = tid / TPR; kk < GEMM_BK16; kk += KPP) {
        const int64_t k = k0 + kk;
        float v[4];
        load4(g.A + k * g.a_cs + (m0 + iq), 0, m0 + iq, g.M, k < k_end, g.a_vec, v);
#pragma unroll
        for (int e = 0; e < 4; ++e) As[(iq + e) * LDK + kk] = f32_to_bf16_rne(v[e]);
      }
    }
    if (B_KC) {
      constexpr int TPR = GEMM_BK16 / 4;
      constexpr int RPP = RX_BLOCK / TPR;
      const int kq = (tid % TPR) * 4;
#pragma unroll
      for (int r = tid / TPR; r < BN; r += RPP) {
        const int64_t col = n0 + r;
        float v[4];
        load4(g.B + col * g.b_cs + (k0 + kq), 0, k0 + kq, k_end, col < g.N, g.b_vec, v);
        ushort4 o = make_ushort4(f32_to_bf16_rne(v[0]), f32_to_bf16_rne(v[1]), f32_to_bf16_rne(v[2]), f32_to_bf16_rne(v[3]));
        *reinterpret_cast<ushort4*>(&Bs[r * LDK + kq]) = o;
      }
    } else {
      constexpr int TPR = BN / 4;
      constexpr int KPP = RX_BLOCK / TPR;
      const int jq = (tid % TPR) * 4;
#pragma unroll
      for (int kk = tid / TPR; kk < GEMM_BK16; kk += KPP) {
        const int64_t k = k0 + kk;
        float v[4];
        load4(g.B + k * g.b_rs + (n0 + jq), 0, n0 + jq, g.N, k < k_end, g.b_vec, v);
#pragma unroll
        for (int e = 0; e < 4; ++e) Bs[(jq + e) * LDK + kk] = f32_to_bf16_rne(v[e]);
      }
    }
    __syncthreads();
    const int h = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int ks = 0; ks < GEMM_BK16; ks += 16) {
      bf16x8 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        a[i] = *reinterpret_cast<const bf16x8*>(&As[((wave_m * TM + i) * 32 + l31) * LDK + ks + 8 * h]);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        b[j] = *reinterpret_cast<const bf16x8*>(&Bs[((wave_n * TN + j) * 32 + l31) * LDK + ks + 8 * h]);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  float* Cb = g.C + (int64_t)blockIdx.z * g.c_slab_stride;
  const int h = lane >> 5, l31 = lane & 31;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int64_t col = n0 + (wave_n * TN + j) * 32 + l31;
    if (col >= g.N) continue;
    const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + (wave_m * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < g.M) {
          float v = acc[i][j][r] + bv;
          if (g.act == 1) v = v > 0.f ? v : 0.f;
          float* dst = Cb + row * g.c_rs + col;
          *dst = g.accumulate ? (*dst + v) : v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------- bf16 operands, pipelined
// The kernel the aligned shapes take (16-byte aligned operands, leading dimensions and
// contiguous extents and reduction lengths that are multiples of 4).
//   * 128x128 tile, 2x2 waves, each wave 2x2 MFMA tiles of v_mfma_f32_32x32x16_bf16
//   * the next K-step's global loads are issued before the current step's MFMAs and land in
//     registers; they are rounded to bf16 (v_cvt_pk_bf16_f32) and written to the other LDS buffer
//     after the MFMAs: one barrier per K-step, global latency behind matrix work
//   * an operand that is contiguous along K is kept [row][32 + 8] and read 16 bytes per lane;
//     an operand that is contiguous along M/N (dY and X of the weight gradient, W of the input
//     gradient) is kept as loaded, [k][128 + 32], and read through ds_read_b64_tr_b16, the
//     transposing LDS read: no transposing stores, no second copy of the activations.
//     Row strides of 80 and 320 bytes keep both kinds of read off shared banks.
//   * workgroups that share an A panel are placed on one XCD so the panel stays in that L2.
#define G2_BK 32
#define G2_LDK (G2_BK + 8)
#define G2_T 128
#define G2_LDR (G2_T + 32)

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
  f32x2_t v = {a, b};
  bf16x2_t r = __builtin_convertvector(v, bf16x2_t);
  return *reinterpret_cast<uint32_t*>(&r);
}

// One operand tile in flight: 16 floats per thread.
struct G2Regs { float4 v[4]; unsigned ok; };   // ok: one bit per 16-byte piece inside the matrix (zeroed at the LDS store)

// K-contiguous operand: tile rows r0.., 32 k from k0; thread = (row tid/4 [+64], 8 k at (tid%4)*8)
__device__ __forceinline__ void g2_load_kc(const float* __restrict__ P, int64_t rs, int64_t r0, int64_t r_lim,
                                           int64_t k0, int64_t k_end, int tid, G2Regs& t) {
  const int64_t k = k0 + (tid & 3) * 8;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int64_t row = r0 + (tid >> 2) + 64 * p;
    // k_end is a multiple of 4 on this path: each 16-byte half is either whole or absent
    const bool ok = row < r_lim && k < k_end, ok2 = row < r_lim && k + 4 < k_end;
    const float* src = P + (ok ? row : 0) * rs + (ok ? k : 0);
    t.v[2 * p] = *reinterpret_cast<const float4*>(src);
    t.v[2 * p + 1] = *reinterpret_cast<const float4*>(src + (ok2 ? 4 : 0));
    if (p == 0) t.ok = 0u;
    t.ok |= ((ok ? 1u : 0u) | (ok2 ? 2u : 0u)) << (2 * p);
  }
}

__device__ __forceinline__ void g2_store_kc(uint16_t* lds, int tid, const G2Regs& t) {
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 va = ((t.ok >> (2 * p)) & 1u) ? t.v[2 * p] : z, vb = ((t.ok >> (2 * p + 1)) & 1u) ? t.v[2 * p + 1] : z;
    uint4 o;
    o.x = pack_bf16(va.x, va.y);
    o.y = pack_bf16(va.z, va.w);
    o.z = pack_bf16(vb.x, vb.y);
    o.w = pack_bf16(vb.z, vb.w);
    *reinterpret_cast<uint4*>(&lds[((tid >> 2) + 64 * p) * G2_LDK + (tid & 3) * 8]) = o;
  }
}

// M/N-contiguous operand: 32 k-rows of 128 contiguous elements from c0; thread = (k-row tid/32 [+8],
// 4 elements at (tid%32)*4)
__device__ __forceinline__ void g2_load_mc(const float* __restrict__ P, int64_t ks, int64_t c0, int64_t c_lim,
                                           int64_t k0, int64_t k_end, int tid, G2Regs& t) {
  const int64_t c = c0 + (tid & 31) * 4;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int64_t k = k0 + (tid >> 5) + 8 * p;
    const bool ok = k < k_end && c < c_lim;
    t.v[p] = *reinterpret_cast<const float4*>(P + (ok ? k : 0) * ks + (ok ? c : 0));
    if (p == 0) t.ok = 0u;
    t.ok |= (ok ? 1u : 0u) << p;
  }
}

__device__ __forceinline__ void g2_store_mc(uint16_t* lds, int tid, const G2Regs& t) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float4 v = ((t.ok >> p) & 1u) ? t.v[p] : make_float4(0.f, 0.f, 0.f, 0.f);
    uint2 o;
    o.x = pack_bf16(v.x, v.y);
    o.y = pack_bf16(v.z, v.w);
    *reinterpret_cast<uint2*>(&lds[((tid >> 5) + 8 * p) * G2_LDR + (tid & 31) * 4]) = o;
  }
}

// MFMA operand (32 rows from `base`, k = ks + 8*(lane>>5) .. +7) out of either LDS image.
template <bool KC>
__device__ __forceinline__ bf16x8 g2_frag(const uint16_t* lds, int base, int ks, int lane) {
  if constexpr (KC) {
    return *reinterpret_cast<const bf16x8*>(&lds[(base + (lane & 31)) * G2_LDK + ks + 8 * (lane >> 5)]);
  } else {
    // per 16-lane group the transposing read takes a 4(k) x 16(column) block: lane 4q+p supplies
    // the address of k-row q, columns 4p..4p+3, and receives column (lane & 15), k-rows 0..3
    const int q = (lane & 15) >> 2, pp = lane & 3;
    const uint16_t* a0 = &lds[(ks + 8 * (lane >> 5) + q) * G2_LDR + base + 16 * ((lane >> 4) & 1) + 4 * pp];
    typedef s16x4 __attribute__((address_space(3))) * lds_ptr;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a0 + 4 * G2_LDR));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
  }
}

template <bool A_KC, bool B_KC, bool ACCUM>
__global__ void __launch_bounds__(RX_BLOCK)
gemm_bf16_pipe_kernel(GemmArgs g, int swizzle) {
  constexpr int A_SZ = A_KC ? G2_T * G2_LDK : G2_BK * G2_LDR;
  constexpr int B_SZ = B_KC ? G2_T * G2_LDK : G2_BK * G2_LDR;
  __shared__ __attribute__((aligned(16))) uint16_t lds[2][A_SZ + B_SZ];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wave_m = wave >> 1, wave_n = wave & 1;
  // XCD-aware order: hardware deals consecutive workgroup ids round-robin over the 8 XCDs;
  // give each XCD a contiguous run of logical tiles (which share A panels)
  int64_t bid = blockIdx.x;
  if (swizzle) bid = (bid % 8) * (gridDim.x / 8) + bid / 8;
  const int64_t tile_m = bid / g.tiles_n;
  const int tile_n = (int)(bid % g.tiles_n);
  const int64_t m0 = tile_m * G2_T;
  const int64_t n0 = (int64_t)tile_n * G2_T;
  const int64_t k_begin = (int64_t)blockIdx.z * g.k_per_slab;
  const int64_t k_end = (k_begin + g.k_per_slab < g.K) ? k_begin + g.k_per_slab : g.K;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  G2Regs ra, rb;
  auto fetch = [&](int64_t k0) {
    if constexpr (A_KC) g2_load_kc(g.A, g.a_rs, m0, g.M, k0, k_end, tid, ra);
    else g2_load_mc(g.A, g.a_cs, m0, g.M, k0, k_end, tid, ra);
    if constexpr (B_KC) g2_load_kc(g.B, g.b_cs, n0, g.N, k0, k_end, tid, rb);
    else g2_load_mc(g.B, g.b_rs, n0, g.N, k0, k_end, tid, rb);
  };
  auto stash = [&](int buf) {
    if constexpr (A_KC) g2_store_kc(lds[buf], tid, ra);
    else g2_store_mc(lds[buf], tid, ra);
    if constexpr (B_KC) g2_store_kc(lds[buf] + A_SZ, tid, rb);
    else g2_store_mc(lds[buf] + A_SZ, tid, rb);
  };

  fetch(k_begin);
  stash(0);
  __syncthreads();
  int cur = 0;
  for (int64_t k0 = k_begin; k0 < k_end; k0 += G2_BK) {
    const bool more = k0 + G2_BK < k_end;
    if (more) fetch(k0 + G2_BK);
    const uint16_t* As = lds[cur];
    const uint16_t* Bs = lds[cur] + A_SZ;
#pragma unroll
    for (int ks = 0; ks < G2_BK; ks += 16) {
      bf16x8 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = g2_frag<A_KC>(As, (wave_m * 2 + i) * 32, ks, lane);
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = g2_frag<B_KC>(Bs, (wave_n * 2 + j) * 32, ks, lane);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (more) stash(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  // epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  float* Cb = g.C + (int64_t)blockIdx.z * g.c_slab_stride;
  const int h = lane >> 5, l31 = lane & 31;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int64_t col = n0 + (wave_n * 2 + j) * 32 + l31;
    if (col >= g.N) continue;
    const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + (wave_m * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < g.M) {
          float v = acc[i][j][r] + bv;
          if (g.act == 1) v = v > 0.f ? v : 0.f;
          float* dst = Cb + row * g.c_rs + col;
          *dst = ACCUM ? (*dst + v) : v;
        }
      }
    }
  }
}

// out[i] = sum over the slabs of slabs[s][i], in a fixed order: four slab lanes per output
// element (lane q adds slabs q, q+4, ... in increasing order), joined as (l0 + l1) + (l2 + l3).
// Small weight matrices are cut into many slabs, so the lanes matter more than the columns.
// A second array (slabs_b, ...) may ride in the same launch: blocks past the first array's share
// reduce it.  Used for a layer's weight-gradient slabs and its bias-gradient partials.
__global__ void __launch_bounds__(RX_BLOCK)
slab_reduce_kernel(const float* __restrict__ slabs, int64_t n, int n_slabs, float* __restrict__ out,
                   const float* __restrict__ slabs_b, int64_t n_b, int n_slabs_b, float* __restrict__ out_b) {
  __shared__ float part[4][64];
  const int e = threadIdx.x & 63, q = threadIdx.x >> 6;
  int64_t blk = blockIdx.x;
  const int64_t blocks_a = (n + 63) / 64;
  if (blk >= blocks_a) {   // block-uniform switch to the second array
    blk -= blocks_a;
    slabs = slabs_b; n = n_b; n_slabs = n_slabs_b; out = out_b;
  }
  const int64_t i = blk * 64 + e;
  float s = 0.f;
  if (i < n) {
    // a thread adds slabs q, q + 4, q + 8, ... in that order; eight loads are in flight ahead of the
    // additions (a tower layer of a few thousand weights has hundreds of slabs, and with four in
    // flight the launch was a chain of dependent round trips)
    int z = q;
    for (; z + 28 < n_slabs; z += 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = slabs[(int64_t)(z + 4 * u) * n + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; z + 12 < n_slabs; z += 16) {
      const float a = slabs[(int64_t)z * n + i], b = slabs[(int64_t)(z + 4) * n + i];
      const float c = slabs[(int64_t)(z + 8) * n + i], d = slabs[(int64_t)(z + 12) * n + i];
      s += a; s += b; s += c; s += d;
    }
    for (; z < n_slabs; z += 4) s += slabs[(int64_t)z * n + i];
  }
  part[q][e] = s;
  __syncthreads();
  if (q == 0 && i < n) out[i] = (part[0][e] + part[1][e]) + (part[2][e] + part[3][e]);
}

// ---------------------------------------------------------------- fp32 operands, pipelined
// Same pipeline as above on v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulation in k
// order): 128x128 tile, 16 k per step, LDS tiles k-major [16][128 + 4] so that the operand read
// (one float per lane, 32 consecutive rows per half-wave) is conflict free.
#define G3_BK 16
#define G3_LD (G2_T + 4)

struct G3Regs { float4 v[2]; unsigned ok; };   // ok: pieces inside the matrix (the rest is zeroed at the LDS store, not behind the load)

// K-contiguous: thread = (row tid/4 [+64], 4 k at (tid%4)*4); M/N-contiguous: (k-row tid/32 [+8], 4 at (tid%32)*4)
__device__ __forceinline__ void g3_load_kc(const float* __restrict__ P, int64_t rs, int64_t r0, int64_t r_lim,
                                           int64_t k0, int64_t k_end, int tid, G3Regs& t) {
  const int64_t k = k0 + (tid & 3) * 4;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int64_t row = r0 + (tid >> 2) + 64 * p;
    const bool ok = row < r_lim && k < k_end;
    t.v[p] = *reinterpret_cast<const float4*>(P + (ok ? row : 0) * rs + (ok ? k : 0));
    if (p == 0) t.ok = 0u;
    t.ok |= (ok ? 1u : 0u) << p;
  }
}

__device__ __forceinline__ void g3_store_kc(float* lds, int tid, const G3Regs& t) {
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const float4 v = ((t.ok >> p) & 1u) ? t.v[p] : make_float4(0.f, 0.f, 0.f, 0.f);
    float* dst = lds + ((tid & 3) * 4) * G3_LD + (tid >> 2) + 64 * p;
    dst[0] = v.x; dst[G3_LD] = v.y; dst[2 * G3_LD] = v.z; dst[3 * G3_LD] = v.w;
  }
}

__device__ __forceinline__ void g3_load_mc(const float* __restrict__ P, int64_t ks, int64_t c0, int64_t c_lim,
                                           int64_t k0, int64_t k_end, int tid, G3Regs& t) {
  const int64_t c = c0 + (tid & 31) * 4;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int64_t k = k0 + (tid >> 5) + 8 * p;
    const bool ok = k < k_end && c < c_lim;
    t.v[p] = *reinterpret_cast<const float4*>(P + (ok ? k : 0) * ks + (ok ? c : 0));
    if (p == 0) t.ok = 0u;
    t.ok |= (ok ? 1u : 0u) << p;
  }
}

__device__ __forceinline__ void g3_store_mc(float* lds, int tid, const G3Regs& t) {
#pragma unroll
  for (int p = 0; p < 2; ++p)
    *reinterpret_cast<float4*>(lds + ((tid >> 5) + 8 * p) * G3_LD + (tid & 31) * 4) = ((t.ok >> p) & 1u) ? t.v[p] : make_float4(0.f, 0.f, 0.f, 0.f);
}

template <bool A_KC, bool B_KC, bool ACCUM>
__global__ void __launch_bounds__(RX_BLOCK, 4)   // 128 registers: four workgroups per CU (34 KB of LDS each)
gemm_f32_pipe_kernel(GemmArgs g, int swizzle) {
  constexpr int T_SZ = G3_BK * G3_LD;
  __shared__ __attribute__((aligned(16))) float lds[2][2 * T_SZ];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wave_m = wave >> 1, wave_n = wave & 1;
  int64_t bid = blockIdx.x;
  if (swizzle) bid = (bid % 8) * (gridDim.x / 8) + bid / 8;
  const int64_t tile_m = bid / g.tiles_n;
  const int tile_n = (int)(bid % g.tiles_n);
  const int64_t m0 = tile_m * G2_T;
  const int64_t n0 = (int64_t)tile_n * G2_T;
  const int64_t k_begin = (int64_t)blockIdx.z * g.k_per_slab;
  const int64_t k_end = (k_begin + g.k_per_slab < g.K) ? k_begin + g.k_per_slab : g.K;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  G3Regs ra, rb;
  auto fetch = [&](int64_t k0) {
    if constexpr (A_KC) g3_load_kc(g.A, g.a_rs, m0, g.M, k0, k_end, tid, ra);
    else g3_load_mc(g.A, g.a_cs, m0, g.M, k0, k_end, tid, ra);
    if constexpr (B_KC) g3_load_kc(g.B, g.b_cs, n0, g.N, k0, k_end, tid, rb);
    else g3_load_mc(g.B, g.b_rs, n0, g.N, k0, k_end, tid, rb);
  };
  float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);   // M-contiguous A: this thread's 4 columns of A, summed over its k-rows
  const bool want_colsum = !A_KC && g.colsum != nullptr && tile_n == 0;   // block-uniform
  auto stash = [&](int buf) {
    if constexpr (A_KC) g3_store_kc(lds[buf], tid, ra);
    else {
      g3_store_mc(lds[buf], tid, ra);
      if (want_colsum) {
#pragma unroll
        for (int p = 0; p < 2; ++p)
          if ((ra.ok >> p) & 1u) { csum.x += ra.v[p].x; csum.y += ra.v[p].y; csum.z += ra.v[p].z; csum.w += ra.v[p].w; }
      }
    }
    if constexpr (B_KC) g3_store_kc(lds[buf] + T_SZ, tid, rb);
    else g3_store_mc(lds[buf] + T_SZ, tid, rb);
  };

  fetch(k_begin);
  stash(0);
  __syncthreads();
  int cur = 0;
  const int h = lane >> 5, l31 = lane & 31;
  for (int64_t k0 = k_begin; k0 < k_end; k0 += G3_BK) {
    const bool more = k0 + G3_BK < k_end;
    if (more) fetch(k0 + G3_BK);
    const float* As = lds[cur];
    const float* Bs = lds[cur] + T_SZ;
#pragma unroll
    for (int kk = 0; kk < G3_BK; kk += 2) {
      float a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = As[(kk + h) * G3_LD + (wave_m * 2 + i) * 32 + l31];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = Bs[(kk + h) * G3_LD + (wave_n * 2 + j) * 32 + l31];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (more) stash(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  if constexpr (!A_KC) {
    if (want_colsum) {
      // the K loop ended with a barrier: the tiles are free.  8 k-row groups hold sums of the same 128 columns
      float* red = &lds[0][0];
      *reinterpret_cast<float4*>(&red[(tid >> 5) * G2_T + (tid & 31) * 4]) = csum;
      __syncthreads();
      if (tid < G2_T && m0 + tid < g.M) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < RX_BLOCK / 32; ++q) t += red[q * G2_T + tid];
        g.colsum[(int64_t)blockIdx.z * g.M + m0 + tid] = t;
      }
    }
  }

  float* Cb = g.C + (int64_t)blockIdx.z * g.c_slab_stride;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int64_t col = n0 + (wave_n * 2 + j) * 32 + l31;
    if (col >= g.N) continue;
    const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + (wave_m * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < g.M) {
          float v = acc[i][j][r] + bv;
          if (g.act == 1) v = v > 0.f ? v : 0.f;
          float* dst = Cb + row * g.c_rs + col;
          *dst = ACCUM ? (*dst + v) : v;
        }
      }
    }
  }
}

// partial[z][j] = sum over the slab's rows of dY[row, j].  A block covers 64 columns with
// 16 row lanes of 16-byte loads (four rows in flight per thread); VEC = false is the scalar
// form for unaligned dY.
#define COLSUM_SLABS 64
template <bool VEC>
__global__ void __launch_bounds__(RX_BLOCK)
colsum_slab_kernel(const float* __restrict__ dY, int64_t ldd, int64_t M, int N, int64_t rows_per_slab,
                   float* __restrict__ partial) {
  __shared__ float red[16][64 + 1];
  const int cq = threadIdx.x & 15;   // column quad
  const int rl = threadIdx.x >> 4;   // row lane
  const int j0 = blockIdx.x * 64 + cq * 4;
  const int64_t lo = (int64_t)blockIdx.z * rows_per_slab;
  const int64_t hi = (lo + rows_per_slab < M) ? lo + rows_per_slab : M;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  if (VEC) {
    if (j0 < N) {  // N % 4 == 0 on this path
      int64_t r = lo + rl;
      for (; r + 48 < hi; r += 64) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(dY + (r + 16 * u) * ldd + j0);
#pragma unroll
        for (int u = 0; u < 4; ++u) { s[0] += v[u].x; s[1] += v[u].y; s[2] += v[u].z; s[3] += v[u].w; }
      }
      for (; r < hi; r += 16) {
        const float4 v = *reinterpret_cast<const float4*>(dY + r * ldd + j0);
        s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
      }
    }
  } else {
    for (int64_t r = lo + rl; r < hi; r += 16)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (j0 + e < N) s[e] += dY[r * ldd + j0 + e];
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[rl][cq * 4 + e] = s[e];
  __syncthreads();
  if (threadIdx.x < 64) {
    const int j = blockIdx.x * 64 + threadIdx.x;
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q][threadIdx.x];
    if (j < N) partial[(int64_t)blockIdx.z * N + j] = t;
  }
}

// Shapes the pipelined bf16 kernel takes: 16-byte loads must be legal wherever it issues them.
template <bool A_KC, bool B_KC>
static bool pipe_ok(const GemmArgs& g) {
  if (!g.a_vec || !g.b_vec) return false;
  // K-contiguous operands are read as two 16-byte halves of 8 k (every slab starts at a multiple
  // of 32); M/N-contiguous ones 4 elements at a time along M/N
  if ((A_KC || B_KC) && (g.K % 4 != 0)) return false;
  if (!A_KC && (g.M % 4 != 0)) return false;
  if (!B_KC && (g.N % 4 != 0)) return false;
  if (g.k_per_slab % G2_BK != 0 && g.k_per_slab < g.K) return false;
  return true;
}

template <bool A_KC, bool B_KC>
static bool pipe_ok_f32(const GemmArgs& g) {
  if (!g.a_vec || !g.b_vec) return false;
  if ((A_KC || B_KC) && (g.K % 4 != 0)) return false;
  if (!A_KC && (g.M % 4 != 0)) return false;
  if (!B_KC && (g.N % 4 != 0)) return false;
  if (g.k_per_slab % G3_BK != 0 && g.k_per_slab < g.K) return false;
  return true;
}

// Whether gemm_launch sends this fp32 weight-gradient product (both operands contiguous along the outputs)
// to the pipelined 128 x 128 kernel.
static bool gemm_pipe_f32_mc(const GemmArgs& g, int bf16) {
  const bool skinny = !bf16 && g.N <= 128 && (g.k_per_slab <= 128 || g.M <= 64);
  return !bf16 && g.N > 64 && !skinny && pipe_ok_f32<false, false>(g);
}

// Whether gemm_launch sends this fp32 product to the deep-step (64 k per trip) tiles of gemm_f32_kernel.
static bool gemm_deep_f32(const GemmArgs& g, int bf16, bool a_kc, bool b_kc) {
  static const bool bk16 = [] { const char* e = getenv("RX_GEMM_BK16"); return e && e[0] == '1'; }();   // A/B timing: keep 16-k steps
  const bool skinny = !bf16 && g.N <= 128 && (g.k_per_slab <= 128 || g.M <= 64);
  // whole 16-byte pieces only: aligned operands, K and the contiguous extents multiples of 4
  const bool whole4 = g.a_vec && g.b_vec && g.K % 4 == 0 && g.k_per_slab % 4 == 0 && (a_kc || g.M % 4 == 0) && (b_kc || g.N % 4 == 0);
  return skinny && !bk16 && g.k_per_slab >= 64 && whole4;
}

template <bool A_KC, bool B_KC>
static int gemm_launch(GemmArgs& g, int n_slabs, int bf16, hipStream_t st) {
  const int64_t tiles_m = rx_cdiv(g.M, 128);
  if (bf16 && g.N > 64 && pipe_ok<A_KC, B_KC>(g)) {
    g.tiles_n = (int)rx_cdiv(g.N, G2_T);
    const int64_t blocks = tiles_m * g.tiles_n;
    RX_REQUIRE(blocks < (1ll << 31), "gemm: grid too large");
    dim3 grid((unsigned)blocks, 1, (unsigned)n_slabs);
    const int swz = (blocks % 8 == 0) ? 1 : 0;
    if (g.accumulate) hipLaunchKernelGGL((gemm_bf16_pipe_kernel<A_KC, B_KC, true>), grid, dim3(RX_BLOCK), 0, st, g, swz);
    else hipLaunchKernelGGL((gemm_bf16_pipe_kernel<A_KC, B_KC, false>), grid, dim3(RX_BLOCK), 0, st, g, swz);
    RX_LAUNCH_CHECK();
    return 0;
  }
  // Short reductions (K <= 128) and short outputs (M <= 64) at N <= 128 are bound by memory and
  // latency, not by MFMA rate: 64-row tiles of the single-buffered kernel put twice as many
  // workgroups on the chip and beat the pipelined 128 x 128 kernel there (measured at M = 40960:
  // grad-X of 128->64 23.7 -> 18.8 us, its grad-W 49 -> 37 us).
  const bool skinny = !bf16 && g.N <= 128 && (g.k_per_slab <= 128 || g.M <= 64);
  if (!bf16 && g.N > 64 && !skinny && pipe_ok_f32<A_KC, B_KC>(g)) {
    g.tiles_n = (int)rx_cdiv(g.N, G2_T);
    const int64_t blocks = tiles_m * g.tiles_n;
    RX_REQUIRE(blocks < (1ll << 31), "gemm: grid too large");
    dim3 grid((unsigned)blocks, 1, (unsigned)n_slabs);
    const int swz = (blocks % 8 == 0) ? 1 : 0;
    if (g.accumulate) hipLaunchKernelGGL((gemm_f32_pipe_kernel<A_KC, B_KC, true>), grid, dim3(RX_BLOCK), 0, st, g, swz);
    else hipLaunchKernelGGL((gemm_f32_pipe_kernel<A_KC, B_KC, false>), grid, dim3(RX_BLOCK), 0, st, g, swz);
    RX_LAUNCH_CHECK();
    return 0;
  }
#define RX_GEMM_GO(BM_, BN_, WM_, WN_)                                                                  \
  do {                                                                                                  \
    g.tiles_n = (int)rx_cdiv(g.N, BN_);                                                                 \
    const int64_t blocks = rx_cdiv(g.M, BM_) * g.tiles_n;                                               \
    RX_REQUIRE(blocks < (1ll << 31), "gemm: grid too large");                                           \
    dim3 grid((unsigned)blocks, 1, (unsigned)n_slabs);                                                  \
    if (bf16)                                                                                           \
      hipLaunchKernelGGL((gemm_bf16_kernel<A_KC, B_KC, BM_, BN_, WM_, WN_>), grid, dim3(RX_BLOCK), 0, st, g); \
    else                                                                                                \
      hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, BM_, BN_, WM_, WN_>), grid, dim3(RX_BLOCK), 0, st, g);  \
  } while (0)
  // skinny fp32 products: the same tiles with 64 k per trip to memory (see gemm_f32_kernel);
  // measured on the ML-1M tower shapes (M = 40960, 128 -> 64): weight-gradient slabs 34.7 -> 30.8 us, forward
  // 15.8 -> 14.6, input gradient 19.0 -> 18.2 (the last two only with the bank-conflict-free pitch of their
  // transposed tiles; with the common pitch they lost 2 us)
  if (gemm_deep_f32(g, bf16, A_KC, B_KC)) {
#define RX_GEMM_DEEP(BM_, BN_, WM_, WN_)                                                               \
  do {                                                                                                  \
    g.tiles_n = (int)rx_cdiv(g.N, BN_);                                                                 \
    const int64_t blocks = rx_cdiv(g.M, BM_) * g.tiles_n;                                               \
    RX_REQUIRE(blocks < (1ll << 31), "gemm: grid too large");                                           \
    hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, BM_, BN_, WM_, WN_, 64>), dim3((unsigned)blocks, 1, (unsigned)n_slabs), dim3(RX_BLOCK), 0, st, g); \
  } while (0)
    if (g.N <= 32) RX_GEMM_DEEP(128, 32, 4, 1);
    else if (g.N <= 64) RX_GEMM_DEEP(64, 64, 2, 2);
    else RX_GEMM_DEEP(64, 128, 2, 2);
#undef RX_GEMM_DEEP
    RX_LAUNCH_CHECK();
    return 0;
  }
  if (g.N <= 32) RX_GEMM_GO(128, 32, 4, 1);
  else if (g.N <= 64 && !bf16) RX_GEMM_GO(64, 64, 2, 2);   // fwd 128->64 at M = 40960: 23.0 -> 15.7 us
  else if (g.N <= 64) RX_GEMM_GO(128, 64, 2, 2);
  else if (skinny) RX_GEMM_GO(64, 128, 2, 2);
  else RX_GEMM_GO(128, 128, 2, 2);
#undef RX_GEMM_GO
  RX_LAUNCH_CHECK();
  return 0;
}

static inline int vec_ok(const float* p, int64_t ld) { return (ld % 4 == 0) && rx_aligned16(p); }


// ---------------------------------------------------------------- one output column (N = 1)
// The last layer of every tower is Linear(K, 1): a matrix-vector product per direction, bound by
// the one streaming pass over X (or dX).  A tile of the MFMA kernels would carry one useful
// column out of 32.  LPR lanes of 16 bytes cover a row (LPR = power of two >= K / 4, K <= 256).
template <int LPR>
__global__ void __launch_bounds__(RX_BLOCK)
gemv_fwd_kernel(const float* __restrict__ X, int64_t ldx, const float* __restrict__ w, const float* __restrict__ b,
                float* __restrict__ Y, int64_t ldy, int64_t M, int K, int act) {
  constexpr int RL = RX_BLOCK / LPR;
  const int cq = threadIdx.x % LPR, rl = threadIdx.x / LPR;
  const int k0 = cq * 4;
  const bool ok = k0 < K;
  const float4 wv = ok ? *reinterpret_cast<const float4*>(w + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float bv = b ? b[0] : 0.f;
  const int kk = ok ? k0 : 0;
  for (int64_t r0 = (int64_t)blockIdx.x * RL * 4; r0 < M; r0 += (int64_t)gridDim.x * RL * 4) {
    float4 x[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t r = r0 + u * RL + rl;
      x[u] = *reinterpret_cast<const float4*>(X + (r < M ? r : 0) * ldx + kk);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float s = ok ? ((x[u].x * wv.x + x[u].y * wv.y) + (x[u].z * wv.z + x[u].w * wv.w)) : 0.f;
#pragma unroll
      for (int d = LPR / 2; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
      const int64_t r = r0 + u * RL + rl;
      if (cq == 0 && r < M) {
        float v = s + bv;
        if (act == 1) v = v > 0.f ? v : 0.f;
        Y[r * ldy] = v;
      }
    }
  }
}

// dX[m, k] (+)= dy[m] * w[k]
__global__ void __launch_bounds__(RX_BLOCK)
gemv_bwd_x_kernel(const float* __restrict__ dY, int64_t ldd, const float* __restrict__ w, float* __restrict__ dX,
                  int64_t ldx, int64_t M, int K, int accumulate) {
  const int kq = K / 4;
  const int64_t total = M * kq;
  const bool small = total < (1ll << 31);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = rx_row_of(i, kq, small);
    const int k0 = (int)(i - m * kq) * 4;
    const float g = dY[m * ldd];
    const float4 wv = *reinterpret_cast<const float4*>(w + k0);
    float4* dst = reinterpret_cast<float4*>(dX + m * ldx + k0);
    float4 v = make_float4(g * wv.x, g * wv.y, g * wv.z, g * wv.w);
    if (accumulate) {
      const float4 o = *dst;
      v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
    }
    *dst = v;
  }
}

// slab z: part_w[z][k] = sum over its rows of dy[m] * X[m, k], part_b[z] = sum of dy[m]; rows in
// a fixed order per lane, lanes joined in a fixed order, so results are reproducible.
template <int LPR>
__global__ void __launch_bounds__(RX_BLOCK)
gemv_bwd_w_kernel(const float* __restrict__ dY, int64_t ldd, const float* __restrict__ X, int64_t ldx, int64_t M, int K,
                  int64_t rows_per_slab, float* __restrict__ part_w, float* __restrict__ part_b) {
  constexpr int RL = RX_BLOCK / LPR;
  __shared__ float red[RL][LPR * 4 + 1];
  __shared__ float redb[RL];
  const int cq = threadIdx.x % LPR, rl = threadIdx.x / LPR;
  const int k0 = cq * 4;
  const bool ok = k0 < K;
  const int kk = ok ? k0 : 0;
  const int64_t lo = (int64_t)blockIdx.x * rows_per_slab;
  const int64_t hi = (lo + rows_per_slab < M) ? lo + rows_per_slab : M;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  float sb = 0.f;
  int64_t r = lo + rl;
  for (; r + 3 * RL < hi; r += 4 * RL) {
    float4 x[4];
    float g[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      x[u] = *reinterpret_cast<const float4*>(X + (r + u * RL) * ldx + kk);
      g[u] = dY[(r + u * RL) * ldd];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      s[0] += g[u] * x[u].x; s[1] += g[u] * x[u].y; s[2] += g[u] * x[u].z; s[3] += g[u] * x[u].w;
      sb += g[u];
    }
  }
  for (; r < hi; r += RL) {
    const float4 x = *reinterpret_cast<const float4*>(X + r * ldx + kk);
    const float g = dY[r * ldd];
    s[0] += g * x.x; s[1] += g * x.y; s[2] += g * x.z; s[3] += g * x.w;
    sb += g;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[rl][k0 + e] = s[e];
  if (cq == 0) redb[rl] = sb;
  __syncthreads();
  if (threadIdx.x < K) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < RL; ++q) t += red[q][threadIdx.x];
    part_w[(int64_t)blockIdx.x * K + threadIdx.x] = t;
  }
  if (threadIdx.x == 0 && part_b) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < RL; ++q) t += redb[q];
    part_b[blockIdx.x] = t;
  }
}

static inline int gemv_lpr(int K) {
  int l = 1;
  while (l * 4 < K) l <<= 1;
  return l;
}

// N = 1 with 16-byte-addressable rows of at most 256 columns
static inline bool gemv_ok(int32_t N, int32_t K, const float* X, int64_t ldx, const float* w, int bf16) {
  return N == 1 && !bf16 && K % 4 == 0 && K >= 4 && K <= 256 && vec_ok(X, ldx) && rx_aligned16(w);
}

#define RX_GEMV_DISPATCH(KERNEL, lpr, ...)                                                    \
  switch (lpr) {                                                                              \
    case 1: hipLaunchKernelGGL((KERNEL<1>), __VA_ARGS__); break;                              \
    case 2: hipLaunchKernelGGL((KERNEL<2>), __VA_ARGS__); break;                              \
    case 4: hipLaunchKernelGGL((KERNEL<4>), __VA_ARGS__); break;                              \
    case 8: hipLaunchKernelGGL((KERNEL<8>), __VA_ARGS__); break;                              \
    case 16: hipLaunchKernelGGL((KERNEL<16>), __VA_ARGS__); break;                            \
    case 32: hipLaunchKernelGGL((KERNEL<32>), __VA_ARGS__); break;                            \
    default: hipLaunchKernelGGL((KERNEL<64>), __VA_ARGS__); break;                            \
  }

extern "C" int rx_linear_fwd(const float* X, int64_t ldx, const float* W, const float* b, float* Y,
                             int64_t ldy, int64_t M, int32_t N, int32_t K, int act, int bf16, int device,
                             void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(M >= 0 && N >= 1 && K >= 1 && ldx >= K && ldy >= N, "rx_linear_fwd: bad shape M=%lld N=%d K=%d", (long long)M, N, K);
  RX_REQUIRE(act == 0 || act == 1, "rx_linear_fwd: unknown activation %d", act);
  if (M == 0) return 0;
  RX_REQUIRE(X && W && Y, "rx_linear_fwd: null argument");
  GemmArgs g{};
  g.A = X; g.a_rs = ldx; g.a_cs = 1;
  g.B = W; g.b_rs = 1; g.b_cs = K;
  g.C = Y; g.c_rs = ldy; g.bias = b;
  g.M = M; g.N = N; g.K = K; g.act = act; g.accumulate = 0;
  g.k_per_slab = K; g.c_slab_stride = 0;
  g.a_vec = vec_ok(X, ldx); g.b_vec = vec_ok(W, K);
  if (gemv_ok(N, K, X, ldx, W, bf16)) {
    const int lpr = gemv_lpr(K);
    const int64_t rows_per_block = (RX_BLOCK / lpr) * 4;
    int64_t grid = rx_cdiv(M, rows_per_block);
    if (grid > 2048) grid = 2048;
    RX_GEMV_DISPATCH(gemv_fwd_kernel, lpr, dim3((unsigned)grid), dim3(RX_BLOCK), 0, (hipStream_t)stream, X, ldx, W, b, Y, ldy, M, (int)K, act);
    RX_LAUNCH_CHECK();
    return 0;
  }
  return gemm_launch<true, true>(g, 1, bf16, (hipStream_t)stream);
}

extern "C" int rx_linear_bwd_x(const float* dY, int64_t ldd, const float* W, float* dX, int64_t ldx,
                               int64_t M, int32_t N, int32_t K, int accumulate, int bf16, int device,
                               void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(M >= 0 && N >= 1 && K >= 1 && ldd >= N && ldx >= K, "rx_linear_bwd_x: bad shape");
  if (M == 0) return 0;
  RX_REQUIRE(dY && W && dX, "rx_linear_bwd_x: null argument");
  GemmArgs g{};
  g.A = dY; g.a_rs = ldd; g.a_cs = 1;     // A(i,n) = dY[i,n]
  g.B = W; g.b_rs = K; g.b_cs = 1;        // B(n,k) = W[n,k]
  g.C = dX; g.c_rs = ldx; g.bias = nullptr;
  g.M = M; g.N = K; g.K = N; g.act = 0; g.accumulate = accumulate;
  g.k_per_slab = N; g.c_slab_stride = 0;
  g.a_vec = vec_ok(dY, ldd); g.b_vec = vec_ok(W, K);
  if (gemv_ok(N, K, dX, ldx, W, bf16)) {
    int64_t grid = rx_cdiv(M * (K / 4), RX_BLOCK);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(gemv_bwd_x_kernel, dim3((unsigned)grid), dim3(RX_BLOCK), 0, (hipStream_t)stream, dY, ldd, W, dX, ldx, M, (int)K,
                       accumulate);
    RX_LAUNCH_CHECK();
    return 0;
  }
  return gemm_launch<true, false>(g, 1, bf16, (hipStream_t)stream);
}

static int bwd_w_slabs(int64_t M, int32_t N, int32_t K) {
  // enough slabs to put >= ~768 blocks (three per CU) on the chip, each slab a multiple of the k-step
  const int64_t tiles = rx_cdiv(N, 128) * rx_cdiv(K, (K <= 32 ? 32 : (K <= 64 ? 64 : 128)));
  int64_t want = rx_cdiv(768, tiles);
  static const int slab_rows = [] { const char* e = getenv("RX_BWDW_SLAB_ROWS"); const int v = e ? atoi(e) : 0; return v >= 64 ? v : 192; }();
  // rows per slab where the row count caps the slab count (one-tile products of the ML-1M towers): 64-row slabs
  // cost more in the slab sum than they gain; with 64-k trips 192 rows (three trips) edge out 128 and 256
  // (NeuMF step 0.271 / 0.267 / 0.270 ms at 128 / 192 / 256); RX_BWDW_SLAB_ROWS overrides
  int64_t max_by_rows = rx_cdiv(M, slab_rows);
  if (want > max_by_rows) want = max_by_rows;
  if (want < 1) want = 1;
  if (want > 1024) want = 1024;
  return (int)want;
}

extern "C" size_t rx_linear_bwd_w_workspace_bytes(int64_t M, int32_t N, int32_t K) {
  if (M <= 0 || N < 1 || K < 1) return 256;
  const int s = bwd_w_slabs(M, N, K);
  const size_t bslabs = (size_t)(s > COLSUM_SLABS ? s : COLSUM_SLABS);   // the N = 1 path keeps one bias partial per weight slab
  return rx_align_up((size_t)s * N * K * sizeof(float), 256) + rx_align_up(bslabs * N * sizeof(float), 256);
}

extern "C" int rx_linear_bwd_w(const float* dY, int64_t ldd, const float* X, int64_t ldx, float* dW,
                               float* db, int64_t M, int32_t N, int32_t K, int bf16, void* ws,
                               size_t ws_bytes, int device, void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(M >= 1 && N >= 1 && K >= 1 && ldd >= N && ldx >= K, "rx_linear_bwd_w: bad shape");
  RX_REQUIRE(dY && X && dW && ws, "rx_linear_bwd_w: null argument");
  RX_REQUIRE(ws_bytes >= rx_linear_bwd_w_workspace_bytes(M, N, K), "rx_linear_bwd_w: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int s = bwd_w_slabs(M, N, K);
  int64_t rows_per_slab = rx_cdiv(M, s);
  rows_per_slab = rx_cdiv(rows_per_slab, 32) * 32;
  const int n_slabs = (int)rx_cdiv(M, rows_per_slab);
  float* slabs = (float*)ws;
  float* bpart = (float*)((char*)ws + rx_align_up((size_t)s * N * K * sizeof(float), 256));
  if (gemv_ok(N, K, X, ldx, dW, bf16)) {
    const int lpr = gemv_lpr(K);
    RX_GEMV_DISPATCH(gemv_bwd_w_kernel, lpr, dim3((unsigned)n_slabs), dim3(RX_BLOCK), 0, st, dY, ldd, X, ldx, M, (int)K, rows_per_slab,
                     slabs, db ? bpart : (float*)nullptr);
    RX_LAUNCH_CHECK();
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)(rx_cdiv((int64_t)K, 64) + (db ? 1 : 0))), dim3(RX_BLOCK), 0, st, slabs,
                       (int64_t)K, n_slabs, dW, (const float*)bpart, (int64_t)(db ? 1 : 0), n_slabs, db);
    RX_LAUNCH_CHECK();
    return 0;
  }
  GemmArgs g{};
  g.A = dY; g.a_rs = 1; g.a_cs = ldd;     // A(n, m) = dY[m, n]
  g.B = X; g.b_rs = ldx; g.b_cs = 1;      // B(m, k) = X[m, k]
  g.C = slabs; g.c_rs = K; g.bias = nullptr;
  g.M = N; g.N = K; g.K = M; g.act = 0; g.accumulate = 0;
  g.k_per_slab = rows_per_slab; g.c_slab_stride = (int64_t)N * K;
  g.a_vec = vec_ok(dY, ldd); g.b_vec = vec_ok(X, ldx);
  // the deep-step kernel sums the columns of the dY tiles it stages: the bias gradient's partials come out of
  // the product's own launch, one per weight-gradient slab
  const bool fused_bias = db && (gemm_deep_f32(g, bf16, false, false) || gemm_pipe_f32_mc(g, bf16));
  if (fused_bias) g.colsum = bpart;
  if (gemm_launch<false, false>(g, n_slabs, bf16, st)) return 1;
  const int64_t nk = (int64_t)N * K;
  if (fused_bias) {
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)(rx_cdiv(nk, 64) + rx_cdiv(N, 64))), dim3(RX_BLOCK), 0, st, slabs, nk,
                       n_slabs, dW, (const float*)bpart, (int64_t)N, n_slabs, db);
    RX_LAUNCH_CHECK();
    return 0;
  }
  if (!db) {
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)rx_cdiv(nk, 64)), dim3(RX_BLOCK), 0, st, slabs, nk, n_slabs, dW,
                       (const float*)nullptr, (int64_t)0, 0, (float*)nullptr);
    RX_LAUNCH_CHECK();
  }
  if (db) {
    // the bias gradient has its own, finer slabs: the column sums are a pure streaming read
    int64_t cs_rows = rx_cdiv(M, COLSUM_SLABS);
    cs_rows = rx_cdiv(cs_rows, 16) * 16;
    const int cs = (int)rx_cdiv(M, cs_rows);
    const dim3 cgrid((unsigned)rx_cdiv(N, 64), 1, (unsigned)cs);
    if (vec_ok(dY, ldd) && N % 4 == 0) hipLaunchKernelGGL(colsum_slab_kernel<true>, cgrid, dim3(RX_BLOCK), 0, st, dY, ldd, M, N, cs_rows, bpart);
    else hipLaunchKernelGGL(colsum_slab_kernel<false>, cgrid, dim3(RX_BLOCK), 0, st, dY, ldd, M, N, cs_rows, bpart);
    RX_LAUNCH_CHECK();
    // one launch sums the weight-gradient slabs and the bias-gradient partials
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)(rx_cdiv(nk, 64) + rx_cdiv(N, 64))), dim3(RX_BLOCK), 0, st, slabs, nk,
                       n_slabs, dW, (const float*)bpart, (int64_t)N, cs, db);
    RX_LAUNCH_CHECK();
  }
  return 0;
}

__global__ void __launch_bounds__(RX_BLOCK)
relu_bwd_kernel(const float* __restrict__ dY, const float* __restrict__ Y, float* __restrict__ dZ, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dZ[i] = Y[i] > 0.f ? dY[i] : 0.f;
}

__global__ void __launch_bounds__(RX_BLOCK)
relu_bwd_scaled_kernel(const float* __restrict__ dY, const float* __restrict__ Y, float* __restrict__ dZ, int64_t n, float scale) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dZ[i] = Y[i] > 0.f ? dY[i] * scale : 0.f;
}

extern "C" int rx_relu_bwd_scaled(const float* dY, const float* Y, float* dZ, int64_t n, float scale, int device, void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(n >= 0, "rx_relu_bwd_scaled: negative size");
  if (n == 0) return 0;
  RX_REQUIRE(dY && Y && dZ, "rx_relu_bwd_scaled: null argument");
  int64_t grid = rx_cdiv(n, RX_BLOCK * 4);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(relu_bwd_scaled_kernel, dim3((unsigned)grid), dim3(RX_BLOCK), 0, (hipStream_t)stream, dY, Y, dZ, n, scale);
  RX_LAUNCH_CHECK();
  return 0;
}

extern "C" int rx_relu_bwd(const float* dY, const float* Y, float* dZ, int64_t n, int device, void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(n >= 0, "rx_relu_bwd: bad n");
  if (n == 0) return 0;
  RX_REQUIRE(dY && Y && dZ, "rx_relu_bwd: null argument");
  int64_t grid = rx_cdiv(n, RX_BLOCK);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(relu_bwd_kernel, dim3((unsigned)grid), dim3(RX_BLOCK), 0, (hipStream_t)stream, dY, Y, dZ, n);
  RX_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- DCN-v2 cross combine
// 2-D elementwise kernels with one row stride per operand (x0 usually lives in a padded
// tower-input buffer).
__global__ void __launch_bounds__(RX_BLOCK)
cross_fwd_kernel(const float* __restrict__ x0, int64_t ld0, const float* __restrict__ u, int64_t ldu,
                 const float* __restrict__ xl, int64_t ldl, float* __restrict__ xn, int64_t ldn,
                 int64_t rows, int cols) {
  const int64_t n = rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cols;
    const int c = (int)(i - r * cols);
    xn[r * ldn + c] = x0[r * ld0 + c] * u[r * ldu + c] + xl[r * ldl + c];
  }
}

__global__ void __launch_bounds__(RX_BLOCK)
cross_bwd_kernel(const float* __restrict__ dxn, int64_t ldd, const float* __restrict__ x0, int64_t ld0,
                 const float* __restrict__ u, int64_t ldu, float* __restrict__ du, int64_t lddu,
                 float* __restrict__ dx0_acc, int64_t ldx0, float* __restrict__ dxl, int64_t ldxl,
                 int64_t rows, int cols, int accumulate, int identity) {
  const int64_t n = rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cols;
    const int c = (int)(i - r * cols);
    const float d = dxn[r * ldd + c];
    du[r * lddu + c] = d * x0[r * ld0 + c];
    float g0 = d * u[r * ldu + c];
    if (identity) g0 += d;                                   // x_l is x0 itself (first cross layer)
    if (accumulate) g0 += dx0_acc[r * ldx0 + c];
    dx0_acc[r * ldx0 + c] = g0;
    if (dxl) dxl[r * ldxl + c] = d;
  }
}

extern "C" int rx_cross_combine_fwd(const float* x0, int64_t ld0, const float* u, int64_t ldu,
                                    const float* xl, int64_t ldl, float* xn, int64_t ldn, int64_t rows,
                                    int32_t cols, int device, void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(rows >= 0 && cols >= 1 && ld0 >= cols && ldu >= cols && ldl >= cols && ldn >= cols, "rx_cross_combine_fwd: bad shape");
  if (rows == 0) return 0;
  RX_REQUIRE(x0 && u && xl && xn, "rx_cross_combine_fwd: null argument");
  int64_t grid = rx_cdiv(rows * cols, RX_BLOCK);
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(cross_fwd_kernel, dim3((unsigned)grid), dim3(RX_BLOCK), 0, (hipStream_t)stream, x0, ld0, u, ldu, xl, ldl, xn, ldn, rows, cols);
  RX_LAUNCH_CHECK();
  return 0;
}

static int cross_combine_bwd_run(const float* dxn, int64_t ldd, const float* x0, int64_t ld0, const float* u,
                                 int64_t ldu, float* du, int64_t lddu, float* dx0_acc, int64_t ldx0, float* dxl,
                                 int64_t ldxl, int64_t rows, int32_t cols, int accumulate, int identity, int device,
                                 void* stream, const char* who) {
  RX_ENTER(device);
  RX_REQUIRE(rows >= 0 && cols >= 1 && ldd >= cols && ld0 >= cols && ldu >= cols && lddu >= cols && ldx0 >= cols &&
                 (!dxl || ldxl >= cols), "%s: bad shape", who);
  if (rows == 0) return 0;
  RX_REQUIRE(dxn && x0 && u && du && dx0_acc, "%s: null argument", who);
  int64_t grid = rx_cdiv(rows * cols, RX_BLOCK);
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(cross_bwd_kernel, dim3((unsigned)grid), dim3(RX_BLOCK), 0, (hipStream_t)stream, dxn, ldd, x0, ld0, u, ldu, du,
                     lddu, dx0_acc, ldx0, dxl, ldxl, rows, cols, accumulate, identity);
  RX_LAUNCH_CHECK();
  return 0;
}

extern "C" int rx_cross_combine_bwd(const float* dxn, int64_t ldd, const float* x0, int64_t ld0,
                                    const float* u, int64_t ldu, float* du, int64_t lddu, float* dx0_acc,
                                    int64_t ldx0, float* dxl, int64_t ldxl, int64_t rows, int32_t cols,
                                    int device, void* stream) {
  RX_REQUIRE(dxl, "rx_cross_combine_bwd: null argument");
  return cross_combine_bwd_run(dxn, ldd, x0, ld0, u, ldu, du, lddu, dx0_acc, ldx0, dxl, ldxl, rows, cols, 1, 0, device, stream,
                               "rx_cross_combine_bwd");
}

extern "C" int rx_cross_combine_bwd_stack(const float* dxn, int64_t ldd, const float* x0, int64_t ld0,
                                          const float* u, int64_t ldu, float* du, int64_t lddu, float* dx0_acc,
                                          int64_t ldx0, float* dxl, int64_t ldxl, int64_t rows, int32_t cols,
                                          int accumulate, int identity, int device, void* stream) {
  return cross_combine_bwd_run(dxn, ldd, x0, ld0, u, ldu, du, lddu, dx0_acc, ldx0, dxl, ldxl, rows, cols, accumulate, identity,
                               device, stream, "rx_cross_combine_bwd_stack");
}

// ================================================================ bf16 activations in HBM
// The tower kept in bf16 between its layers: an activation (or a pre-activation gradient) that only
// ever feeds matrix products is stored as the bf16 value those products would round it to anyway,
// so the results are the ones of the fp32-storage path above while every operand moves half the
// bytes and no conversion sits between the global load and LDS.  Weights stay fp32 masters; a
// layer's bf16 copy ([N, K], and [K, N] for the input gradient, which makes that product
// K-contiguous on both sides like the forward) is made per use by rx_to_bf16.
//   * 128 x 128 tile, 2 x 2 waves of 2 x 2 v_mfma_f32_32x32x16_bf16, H_BK k per step; next step's
//     16-byte loads in flight behind the MFMAs (register staging, two LDS buffers, one barrier per step)
//   * K-contiguous operands [row][H_BK + 8] read 16 bytes per lane; the weight gradient's operands
//     (contiguous along the output dimensions) kept as loaded [k][128 + 32] and read through the
//     transposing LDS read
//   * epilogue variants as separate instantiations: fp32 or bf16 output, bias + ReLU, accumulate
//     into fp32, and the ReLU gate of the layer in front applied from its stored bf16 output
#ifndef H_BK
#define H_BK 32   // 64 was measured too: 72 KB of operand buffers leave 2 workgroups per CU; 40 KB leave 3, which pays (DCN-v2 step 3.02 -> 2.89 ms)
#endif
#define H_LDK (H_BK + 8)
#define H_CPR (H_BK / 8)        // 16-byte pieces per K-contiguous tile row
#define H_NP (H_BK / 16)        // pieces per thread and operand
#define H_T 128
#define H_LDR (H_T + 32)
#define H_LDC (H_T + 4)   // fp32 image of the output tile: 16 consecutive rows of one 16-byte column piece cover all 64 banks

struct Gemm16Args {
  const uint16_t* A; int64_t lda;   // K-contiguous: A(i,k) = A[i*lda + k]; else A(i,k) = A[k*lda + i]
  const uint16_t* B; int64_t ldb;   // K-contiguous: B(k,j) = B[j*ldb + k]; else B(k,j) = B[k*ldb + j]
  void* C; int64_t ldc;             // [M, N] fp32 or bf16
  const float* bias;                // [N] or null
  const uint16_t* gate; int64_t ldg;  // [M, N] bf16: output forced to 0 where gate <= 0
  int64_t M, N, K;
  int act;
  int64_t k_per_slab, c_slab_stride;
  int tiles_n;
  // cross-layer epilogue: with u = A B + bias written to C, also x_next = x0 * u + xl (fp32 and bf16)
  const float* x0; int64_t ld0;
  const float* xl; int64_t ldl;
  float* xn; int64_t ldn;
  uint16_t* xn16; int64_t ldn16;
  // weight-gradient products (operands contiguous along the outputs): [slab][M] column sums of A over the
  // slab's k range, formed from the A tiles the kernel stages (the layer's bias gradient), or null
  float* colsum;
};

// Pieces in flight between a K-step's global loads and its LDS stores.  Out-of-range pieces are
// loaded from a clamped (valid) address and zeroed only when they are written to LDS: a select
// right behind the load would make the wave wait for the load before the MFMAs it is meant to hide behind.
template <int NP = H_NP> struct HRegs { uint4 v[NP]; unsigned ok; };

// K-contiguous: thread = (row tid/8 [+32p], 8 k at (tid%8)*8); K % 8 == 0, so a 16-byte piece is whole or absent
template <int NP>
__device__ __forceinline__ void h_load_kc(const uint16_t* __restrict__ P, int64_t ld, int64_t r0, int64_t r_lim,
                                          int64_t k0, int64_t k_end, int tid, HRegs<NP>& t) {
  const int64_t k = k0 + (tid % H_CPR) * 8;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int64_t row = r0 + (tid / H_CPR) + (RX_BLOCK / H_CPR) * p;
    const bool ok = row < r_lim && k < k_end;
    t.v[p] = *reinterpret_cast<const uint4*>(P + (ok ? row : 0) * ld + (ok ? k : 0));
    if (p == 0) t.ok = 0u;
    t.ok |= (ok ? 1u : 0u) << p;
  }
}

template <int NP>
__device__ __forceinline__ void h_store_kc(uint16_t* lds, int tid, const HRegs<NP>& t) {
#pragma unroll
  for (int p = 0; p < NP; ++p)
    *reinterpret_cast<uint4*>(&lds[((tid / H_CPR) + (RX_BLOCK / H_CPR) * p) * H_LDK + (tid % H_CPR) * 8]) = ((t.ok >> p) & 1u) ? t.v[p] : make_uint4(0u, 0u, 0u, 0u);
}

// contiguous along the output dimension: 64 k-rows of 128 elements; thread = (k-row tid/16 [+16p], 8 at (tid%16)*8)
__device__ __forceinline__ void h_load_mc(const uint16_t* __restrict__ P, int64_t ld, int64_t c0, int64_t c_lim,
                                          int64_t k0, int64_t k_end, int tid, HRegs<>& t) {
  const int64_t c = c0 + (tid & 15) * 8;
#pragma unroll
  for (int p = 0; p < H_NP; ++p) {
    const int64_t k = k0 + (tid >> 4) + 16 * p;
    const bool ok = k < k_end && c < c_lim;
    t.v[p] = *reinterpret_cast<const uint4*>(P + (ok ? k : 0) * ld + (ok ? c : 0));
    if (p == 0) t.ok = 0u;
    t.ok |= (ok ? 1u : 0u) << p;
  }
}

__device__ __forceinline__ void h_store_mc(uint16_t* lds, int tid, const HRegs<>& t) {
#pragma unroll
  for (int p = 0; p < H_NP; ++p)
    *reinterpret_cast<uint4*>(&lds[((tid >> 4) + 16 * p) * H_LDR + (tid & 15) * 8]) = ((t.ok >> p) & 1u) ? t.v[p] : make_uint4(0u, 0u, 0u, 0u);
}

template <bool KC>
__device__ __forceinline__ bf16x8 h_frag(const uint16_t* lds, int base, int ks, int lane) {
  if constexpr (KC) {
    return *reinterpret_cast<const bf16x8*>(&lds[(base + (lane & 31)) * H_LDK + ks + 8 * (lane >> 5)]);
  } else {
    const int q = (lane & 15) >> 2, pp = lane & 3;
    const uint16_t* a0 = &lds[(ks + 8 * (lane >> 5) + q) * H_LDR + base + 16 * ((lane >> 4) & 1) + 4 * pp];
    typedef s16x4 __attribute__((address_space(3))) * lds_ptr;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a0 + 4 * H_LDR));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
  }
}

// KC: both operands K-contiguous (forward, input gradient); !KC: both contiguous along the output
// dimensions (weight gradient, K = batch, slabs over blockIdx.z, fp32 output only)
// One k-step of a wavefront's products out of the LDS operand tiles.
template <bool KC, int WM>
__device__ __forceinline__ void h_multiply(const uint16_t* As, const uint16_t* Bs, int wave_m, int wave_n, int lane, f32x16 (&acc)[WM][2]) {
#pragma unroll
  for (int ks = 0; ks < H_BK; ks += 16) {
    bf16x8 a[WM], b[2];
#pragma unroll
    for (int i = 0; i < WM; ++i) a[i] = h_frag<KC>(As, (wave_m * WM + i) * 32, ks, lane);
#pragma unroll
    for (int j = 0; j < 2; ++j) b[j] = h_frag<KC>(Bs, (wave_n * 2 + j) * 32, ks, lane);
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);   // C^T tile: see the epilogue
  }
}

// BM: rows of the output tile, 128 or 256 (K-contiguous operands only).  With 256 rows a wavefront owns
// 128 x 64 outputs (4 x 2 products per 16-wide k-step): six 1 KB operand reads from LDS feed eight
// products instead of four feeding four.
template <bool KC, bool OUT16, bool GATE, bool ACCUM, bool CROSS = false, int BM = H_T>
__global__ void __launch_bounds__(RX_BLOCK, BM == H_T ? 3 : 2)   // workgroups per CU the register budget must allow
gemm_h16_kernel(Gemm16Args g, int swizzle) {
  static_assert(BM == H_T || (KC && BM == 2 * H_T), "256-row tiles for K-contiguous operands only");
  constexpr int WM = BM / 64;                  // 32-row products per wavefront along M
  constexpr int NPA = H_NP * (BM / H_T);       // 16-byte pieces of A per thread and k-step
  constexpr int A_SZ = KC ? BM * H_LDK : H_BK * H_LDR;
  constexpr int B_SZ = KC ? H_T * H_LDK : H_BK * H_LDR;
  constexpr int OP_SZ = A_SZ + B_SZ;
  extern __shared__ __attribute__((aligned(16))) uint16_t h_lds[];   // [2][A_SZ + B_SZ]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wave_m = wave >> 1, wave_n = wave & 1;
  int64_t bid = blockIdx.x;
  if (swizzle) bid = (bid % 8) * (gridDim.x / 8) + bid / 8;
  const int64_t tile_m = bid / g.tiles_n;
  const int tile_n = (int)(bid % g.tiles_n);
  const int64_t m0 = tile_m * BM;
  const int64_t n0 = (int64_t)tile_n * H_T;
  const int64_t k_begin = (int64_t)blockIdx.z * g.k_per_slab;
  const int64_t k_end = (k_begin + g.k_per_slab < g.K) ? k_begin + g.k_per_slab : g.K;

  f32x16 acc[WM][2];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // the bias entries of this thread's epilogue columns, fetched now: behind the K loop their latency
  // would sit between the tile's last MFMA and its first store
  constexpr bool BIAS = KC && !GATE && !ACCUM;   // the instantiations a forward product can take
  float bias_r[BIAS ? (OUT16 ? 8 : 4) : 1];
  if constexpr (BIAS) {
    const int64_t col = n0 + (OUT16 ? (tid & 15) * 8 : (tid & 31) * 4);
#pragma unroll
    for (int e = 0; e < (OUT16 ? 8 : 4); ++e) bias_r[e] = (g.bias && col + e < g.N) ? g.bias[col + e] : 0.f;
  }

  // Operand pieces travel global -> registers -> LDS one k-step ahead of the products.  (Two steps
  // ahead -- a second register set, the loop written out for two steps -- was measured: no gain at any
  // of the tower shapes, and the extra registers cost the 128-row tiles a workgroup per CU.)
  HRegs<NPA> ra;
  HRegs<> rb;
  auto fetch = [&](int64_t k0) {
    if constexpr (KC) {
      h_load_kc(g.A, g.lda, m0, g.M, k0, k_end, tid, ra);
      h_load_kc(g.B, g.ldb, n0, g.N, k0, k_end, tid, rb);
    } else {
      h_load_mc(g.A, g.lda, m0, g.M, k0, k_end, tid, ra);
      h_load_mc(g.B, g.ldb, n0, g.N, k0, k_end, tid, rb);
    }
  };
  float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // !KC: this thread's 8 columns of A, summed over its k-rows
  const bool want_colsum = !KC && g.colsum != nullptr && tile_n == 0;   // block-uniform
  auto stash = [&](int buf) {
    uint16_t* base = h_lds + buf * OP_SZ;
    if constexpr (KC) { h_store_kc(base, tid, ra); h_store_kc(base + A_SZ, tid, rb); }
    else {
      h_store_mc(base, tid, ra); h_store_mc(base + A_SZ, tid, rb);
      if (want_colsum) {
#pragma unroll
        for (int p = 0; p < NPA; ++p) {
          if (!((ra.ok >> p) & 1u)) continue;
          const uint32_t xs[4] = {ra.v[p].x, ra.v[p].y, ra.v[p].z, ra.v[p].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) { csum[2 * e] += bf16_bits_to_f32(xs[e] & 0xffffu); csum[2 * e + 1] += bf16_bits_to_f32(xs[e] >> 16); }
        }
      }
    }
  };

  fetch(k_begin);
  stash(0);
  __syncthreads();
  int cur = 0;
  for (int64_t k0 = k_begin; k0 < k_end; k0 += H_BK) {
    const bool more = k0 + H_BK < k_end;
    if (more) fetch(k0 + H_BK);
    h_multiply<KC, WM>(h_lds + cur * OP_SZ, h_lds + cur * OP_SZ + A_SZ, wave_m, wave_n, lane, acc);
    if (more) stash(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  if constexpr (!KC) {
    if (want_colsum) {
      // the K loop ended with a barrier: the operand buffers are free.  16 k-row groups hold sums of the same
      // 128 columns; they are added in group order
      float* red = reinterpret_cast<float*>(h_lds);
#pragma unroll
      for (int e = 0; e < 8; ++e) red[(tid >> 4) * H_T + (tid & 15) * 8 + e] = csum[e];
      __syncthreads();
      if (tid < H_T && m0 + tid < g.M) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < RX_BLOCK / 16; ++q) t += red[q * H_T + tid];
        g.colsum[(int64_t)blockIdx.z * g.M + m0 + tid] = t;
      }
      __syncthreads();
    }
  }

  // epilogue: the products were issued with the operands swapped, so an accumulator tile is C^T in the
  // C/D layout: a lane holds ONE output row (lane & 31) and, per group q = r >> 2, four consecutive
  // output columns 8 q + 4 (lane >> 5) + (r & 3).  They go through an fp32 LDS image of the tile with
  // 16-byte writes, so that the output, the gate and the accumulate operand are all touched as whole
  // 16-byte pieces of contiguous rows
  float* ct = reinterpret_cast<float*>(h_lds);
  const int h = lane >> 5, l31 = lane & 31;
  // 64 rows at a time (the image then fits beside nothing bigger than the operand buffers): pass
  // (wm, hp) holds the rows of wavefront row wm's products 2 hp and 2 hp + 1
#pragma unroll
  for (int hp = 0; hp < WM / 2; ++hp) {
#pragma unroll 1
  for (int wm = 0; wm < 2; ++wm) {
    if (hp || wm) __syncthreads();
    if (wave_m == wm) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(&ct[(i * 32 + l31) * H_LDC + (wave_n * 2 + j) * 32 + 8 * q + 4 * h]) =
                make_float4(acc[2 * hp + i][j][4 * q], acc[2 * hp + i][j][4 * q + 1], acc[2 * hp + i][j][4 * q + 2], acc[2 * hp + i][j][4 * q + 3]);
    }
    __syncthreads();
    const int64_t mh = m0 + 64 * (wm * (WM / 2) + hp);
    if constexpr (OUT16) {
      // 8 columns per thread: 16 threads per row, 16 rows per pass
      const int c8 = (tid & 15) * 8;
      const int64_t col = n0 + c8;
      if (col < g.N) {   // a piece that straddles N ends in the row's zero padding (leading dimensions are multiples of 8)
        float bv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) bv[e] = BIAS ? bias_r[BIAS ? e : 0] : 0.f;
        uint16_t* C16 = reinterpret_cast<uint16_t*>(g.C);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int rr = (tid >> 4) + 16 * p;
          const int64_t row = mh + rr;
          if (row >= g.M) continue;
          const float4 lo = *reinterpret_cast<const float4*>(&ct[rr * H_LDC + c8]);
          const float4 hi = *reinterpret_cast<const float4*>(&ct[rr * H_LDC + c8 + 4]);
          float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
          uint32_t gs[4] = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
          if constexpr (GATE) {
            const uint4 gt = *reinterpret_cast<const uint4*>(g.gate + row * g.ldg + col);
            gs[0] = gt.x; gs[1] = gt.y; gs[2] = gt.z; gs[3] = gt.w;
          }
          uint32_t o[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float a = v[2 * e] + bv[2 * e], c = v[2 * e + 1] + bv[2 * e + 1];
            if (g.act == 1) { a = a > 0.f ? a : 0.f; c = c > 0.f ? c : 0.f; }
            if constexpr (GATE) {
              const uint32_t glo = gs[e] & 0xffffu, ghi = gs[e] >> 16;
              if (glo == 0 || (glo & 0x8000u)) a = 0.f;
              if (ghi == 0 || (ghi & 0x8000u)) c = 0.f;
            }
            o[e] = (uint32_t)f32_to_bf16_rne(a) | ((uint32_t)f32_to_bf16_rne(c) << 16);
          }
          *reinterpret_cast<uint4*>(C16 + row * g.ldc + col) = make_uint4(o[0], o[1], o[2], o[3]);
        }
      }
    } else {
      // 4 columns per thread: 32 threads per row, 8 rows per pass.  ldc % 4 == 0 here; a weight-gradient
      // slab has N % 8 == 0, the other fp32 outputs N % 4 == 0 (checked by the callers)
      const int c4 = (tid & 31) * 4;
      const int64_t col = n0 + c4;
      if (col < g.N) {
        const float4 bv = BIAS ? make_float4(bias_r[0], bias_r[BIAS ? 1 : 0], bias_r[BIAS ? 2 : 0], bias_r[BIAS ? 3 : 0]) : make_float4(0.f, 0.f, 0.f, 0.f);
        float* Cf = reinterpret_cast<float*>(g.C) + (int64_t)blockIdx.z * g.c_slab_stride;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
          const int rr = (tid >> 5) + 8 * p;
          const int64_t row = mh + rr;
          if (row >= g.M) continue;
          float4 v = *reinterpret_cast<const float4*>(&ct[rr * H_LDC + c4]);
          v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
          if (g.act == 1) { v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f; }
          float4* dst = reinterpret_cast<float4*>(Cf + row * g.ldc + col);
          if constexpr (ACCUM) { const float4 o = *dst; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
          *dst = v;
          if constexpr (CROSS) {   // the DCN-v2 combine on the tile just produced: x_next = x0 * u + x_l
            const float4 a = *reinterpret_cast<const float4*>(g.x0 + row * g.ld0 + col);
            const float4 d = *reinterpret_cast<const float4*>(g.xl + row * g.ldl + col);
            const float4 o = make_float4(a.x * v.x + d.x, a.y * v.y + d.y, a.z * v.z + d.z, a.w * v.w + d.w);
            *reinterpret_cast<float4*>(g.xn + row * g.ldn + col) = o;
            *reinterpret_cast<uint2*>(g.xn16 + row * g.ldn16 + col) =
                make_uint2((uint32_t)f32_to_bf16_rne(o.x) | ((uint32_t)f32_to_bf16_rne(o.y) << 16),
                           (uint32_t)f32_to_bf16_rne(o.z) | ((uint32_t)f32_to_bf16_rne(o.w) << 16));
          }
        }
      }
    }
  }
  }
}

#define H_LDS_OPERANDS(KC_, BM_) ((size_t)2 * ((KC_) ? ((BM_) + H_T) * H_LDK : 2 * H_BK * H_LDR) * sizeof(uint16_t))
#define H_LDS_IMAGE ((size_t)64 * H_LDC * sizeof(float))
#define H_LDS_BYTES(KC_, BM_) (H_LDS_OPERANDS(KC_, BM_) > H_LDS_IMAGE ? H_LDS_OPERANDS(KC_, BM_) : H_LDS_IMAGE)

template <bool KC, bool OUT16, bool GATE, bool ACCUM, bool CROSS = false, int BM = H_T>
static int gemm_h16_launch_bm(Gemm16Args& g, int n_slabs, hipStream_t st);

// Tall tiles (256 rows) where the grid still fills the card a few times over; RX_GEMM16_TILE_ROWS=128
// in the environment keeps every product on 128-row tiles (A/B timing).
static bool h16_tall_tiles(const Gemm16Args& g) {
  static const int forced = [] { const char* e = getenv("RX_GEMM16_TILE_ROWS"); return e ? atoi(e) : 0; }();
  if (forced == 128) return false;
  return rx_cdiv(g.M, 2 * H_T) * rx_cdiv(g.N, H_T) >= 512;
}

template <bool KC, bool OUT16, bool GATE, bool ACCUM, bool CROSS = false>
static int gemm_h16_launch(Gemm16Args& g, int n_slabs, hipStream_t st) {
  // measured at the tower shapes (M = 65536): tall tiles pay for plain forward products (fp32 out 10-15 %,
  // bf16 out 1-5 %); the gated and accumulating input-gradient products, bound by their extra operand, lose
  if constexpr (KC && !GATE && !ACCUM) {
    if (h16_tall_tiles(g)) return gemm_h16_launch_bm<KC, OUT16, GATE, ACCUM, CROSS, 2 * H_T>(g, n_slabs, st);
  }
  return gemm_h16_launch_bm<KC, OUT16, GATE, ACCUM, CROSS, H_T>(g, n_slabs, st);
}

template <bool KC, bool OUT16, bool GATE, bool ACCUM, bool CROSS, int BM>
static int gemm_h16_launch_bm(Gemm16Args& g, int n_slabs, hipStream_t st) {
  g.tiles_n = (int)rx_cdiv(g.N, H_T);
  const int64_t blocks = rx_cdiv(g.M, BM) * g.tiles_n;
  RX_REQUIRE(blocks < (1ll << 31), "gemm16: grid too large");
  const size_t lds = H_LDS_BYTES(KC, BM);
  // 72 KB / 80 KB of dynamic LDS need the attribute once per instantiation and device
  static bool raised[64] = {};
  int device = 0;
  RX_HIP(hipGetDevice(&device));
  if (device < 0 || device >= 64 || !raised[device]) {
    RX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_h16_kernel<KC, OUT16, GATE, ACCUM, CROSS, BM>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (device >= 0 && device < 64) raised[device] = true;
  }
  const int swz = (blocks % 8 == 0) ? 1 : 0;
  hipLaunchKernelGGL((gemm_h16_kernel<KC, OUT16, GATE, ACCUM, CROSS, BM>), dim3((unsigned)blocks, 1, (unsigned)n_slabs), dim3(RX_BLOCK), lds, st, g, swz);
  RX_LAUNCH_CHECK();
  return 0;
}

static inline bool h16_ok(const void* p, int64_t ld) { return ld % 8 == 0 && rx_aligned16(p); }
static inline bool f4_ok(const void* p, int64_t ld) { return ld % 4 == 0 && rx_aligned16(p); }

// ---- fp32 -> bf16 copies (activations entering the tower, per-use weight copies, optionally transposed)
__global__ void __launch_bounds__(RX_BLOCK)
to_bf16_kernel(const float* __restrict__ src, int64_t lds_, uint16_t* __restrict__ dst, int64_t ldd, int64_t rows, int cols) {
  const int cq = (cols + 3) / 4;
  const int64_t total = rows * cq;
  const bool small = total < (1ll << 31);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = rx_row_of(i, cq, small);
    const int c = (int)(i - r * cq) * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (c + e < cols) dst[r * ldd + c + e] = f32_to_bf16_rne(src[r * lds_ + c + e]);
  }
}

// dst[c, r] = bf16(src[r, c]) through a 32 x 32 LDS tile (weights only: small)
__global__ void __launch_bounds__(RX_BLOCK)
to_bf16_t_kernel(const float* __restrict__ src, int64_t lds_, uint16_t* __restrict__ dst, int64_t ldd, int rows, int cols) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = r0 + ty + 8 * q, c = c0 + tx;
    tile[ty + 8 * q][tx] = (r < rows && c < cols) ? src[(int64_t)r * lds_ + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = c0 + ty + 8 * q, r = r0 + tx;
    if (c < cols && r < rows) dst[(int64_t)c * ldd + r] = f32_to_bf16_rne(tile[tx][ty + 8 * q]);
  }
}

extern "C" int rx_to_bf16(const float* src, int64_t ld_src, uint16_t* dst, int64_t ld_dst, int64_t rows, int32_t cols,
                          int transpose, int device, void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(rows >= 0 && cols >= 1 && ld_src >= cols && ld_dst >= (transpose ? rows : cols), "rx_to_bf16: bad shape");
  if (rows == 0) return 0;
  RX_REQUIRE(src && dst, "rx_to_bf16: null argument");
  if (transpose) {
    RX_REQUIRE(rows < (1 << 20), "rx_to_bf16: the transposing copy is for weight matrices (rows < 2^20)");
    hipLaunchKernelGGL(to_bf16_t_kernel, dim3((unsigned)rx_cdiv(cols, 32), (unsigned)rx_cdiv(rows, 32)), dim3(RX_BLOCK), 0, (hipStream_t)stream,
                       src, ld_src, dst, ld_dst, (int)rows, (int)cols);
  } else {
    int64_t grid = rx_cdiv(rows * ((cols + 3) / 4), RX_BLOCK);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(to_bf16_kernel, dim3((unsigned)grid), dim3(RX_BLOCK), 0, (hipStream_t)stream, src, ld_src, dst, ld_dst, rows, (int)cols);
  }
  RX_LAUNCH_CHECK();
  return 0;
}

// ---- one output column (the tower's last layer) with bf16 activations: K / 8 lanes of 16 bytes per row

template <int LPR>
__global__ void __launch_bounds__(RX_BLOCK)
gemv16_fwd_kernel(const uint16_t* __restrict__ X, int64_t ldx, const float* __restrict__ w, const float* __restrict__ b,
                  float* __restrict__ Y, int64_t ldy, int64_t M, int K, int act) {
  constexpr int RL = RX_BLOCK / LPR;
  const int cq = threadIdx.x % LPR, rl = threadIdx.x / LPR;
  const int k0 = cq * 8;
  const bool ok = k0 < K;
  float wv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) wv[e] = (ok && k0 + e < K) ? round_bf16(w[k0 + e]) : 0.f;   // rows are zero padded to 8
  const float bv = b ? b[0] : 0.f;
  for (int64_t r = (int64_t)blockIdx.x * RL + rl; r < M; r += (int64_t)gridDim.x * RL) {
    const uint4 x = ok ? *reinterpret_cast<const uint4*>(X + r * ldx + k0) : make_uint4(0u, 0u, 0u, 0u);
    const uint32_t xs[4] = {x.x, x.y, x.z, x.w};
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) s += bf16_bits_to_f32(xs[e] & 0xffffu) * wv[2 * e] + bf16_bits_to_f32(xs[e] >> 16) * wv[2 * e + 1];
#pragma unroll
    for (int d = LPR / 2; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if (cq == 0) {
      float v = s + bv;
      if (act == 1) v = v > 0.f ? v : 0.f;
      Y[r * ldy] = v;
    }
  }
}

// dZ16[m, k] = gate(h16[m, k]) * bf16(dy16[m] * bf16(w[k]))  (gate may be null); OUT32: the products
// themselves in fp32 (exact: two bf16 factors), for a consumer that is not a matrix product
template <bool OUT32>
__global__ void __launch_bounds__(RX_BLOCK)
gemv16_bwd_x_kernel(const uint16_t* __restrict__ dY, int64_t ldd, const float* __restrict__ w, const uint16_t* __restrict__ gate,
                    int64_t ldg, void* __restrict__ dZ_, int64_t ldz, int64_t M, int K) {
  uint16_t* dZ = reinterpret_cast<uint16_t*>(dZ_);
  const int kq = (K + 7) / 8;
  const int64_t total = M * kq;
  const bool small = total < (1ll << 31);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = rx_row_of(i, kq, small);
    const int k0 = (int)(i - m * kq) * 8;
    const float gy = bf16_bits_to_f32(dY[m * ldd]);
    float wr[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) wr[e] = k0 + e < K ? round_bf16(w[k0 + e]) : 0.f;
    uint4 gt = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
    if (gate) gt = *reinterpret_cast<const uint4*>(gate + m * ldg + k0);
    const uint32_t gs[4] = {gt.x, gt.y, gt.z, gt.w};
    uint32_t o[4];
    float f[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t lo = gs[e] & 0xffffu, hi = gs[e] >> 16;
      const float a = (lo != 0 && !(lo & 0x8000u)) ? gy * wr[2 * e] : 0.f;
      const float c = (hi != 0 && !(hi & 0x8000u)) ? gy * wr[2 * e + 1] : 0.f;
      f[2 * e] = a; f[2 * e + 1] = c;
      o[e] = (uint32_t)f32_to_bf16_rne(a) | ((uint32_t)f32_to_bf16_rne(c) << 16);
    }
    if constexpr (OUT32) {
      float* dst = reinterpret_cast<float*>(dZ_) + m * ldz + k0;
      *reinterpret_cast<float4*>(dst) = make_float4(f[0], f[1], f[2], f[3]);
      if (k0 + 4 < K) *reinterpret_cast<float4*>(dst + 4) = make_float4(f[4], f[5], f[6], f[7]);   // K % 4 == 0
    } else {
      *reinterpret_cast<uint4*>(dZ + m * ldz + k0) = make_uint4(o[0], o[1], o[2], o[3]);
    }
  }
}

// slab z: part_w[z][k] = sum_m dy16[m] * X16[m, k], part_b[z] = sum_m dy16[m]
template <int LPR>
__global__ void __launch_bounds__(RX_BLOCK)
gemv16_bwd_w_kernel(const uint16_t* __restrict__ dY, int64_t ldd, const uint16_t* __restrict__ X, int64_t ldx, int64_t M, int K,
                    int64_t rows_per_slab, float* __restrict__ part_w, float* __restrict__ part_b) {
  constexpr int RL = RX_BLOCK / LPR;
  __shared__ float red[RL][LPR * 8 + 1];
  __shared__ float redb[RL];
  const int cq = threadIdx.x % LPR, rl = threadIdx.x / LPR;
  const int k0 = cq * 8;
  const bool ok = k0 < K;
  const int64_t lo = (int64_t)blockIdx.x * rows_per_slab;
  const int64_t hi = (lo + rows_per_slab < M) ? lo + rows_per_slab : M;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float sb = 0.f;
  for (int64_t r = lo + rl; r < hi; r += RL) {
    const uint4 x = ok ? *reinterpret_cast<const uint4*>(X + r * ldx + k0) : make_uint4(0u, 0u, 0u, 0u);
    const float gy = bf16_bits_to_f32(dY[r * ldd]);
    const uint32_t xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s[2 * e] += gy * bf16_bits_to_f32(xs[e] & 0xffffu);
      s[2 * e + 1] += gy * bf16_bits_to_f32(xs[e] >> 16);
    }
    sb += gy;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[rl][k0 + e] = s[e];
  if (cq == 0) redb[rl] = sb;
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += RX_BLOCK) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < RL; ++q) t += red[q][k];
    part_w[(int64_t)blockIdx.x * K + k] = t;
  }
  if (threadIdx.x == 0 && part_b) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < RL; ++q) t += redb[q];
    part_b[blockIdx.x] = t;
  }
}

#define GEMV16_MAX_K 4096
static inline int gemv16_lpr(int K) {
  int l = 1;
  while (l * 8 < K) l <<= 1;
  return l;
}
static inline bool gemv16_ok(int32_t N, int32_t K) { return N == 1 && K % 4 == 0 && K >= 8 && K <= GEMV16_MAX_K; }
static inline bool gemv16_wide(int32_t K) { return K > 512; }   // beyond one 16-byte piece per lane of a wavefront

// The same three products for rows of more than 512 inputs (up to GEMV16_MAX_K): a wavefront per row, CH 16-byte
// pieces per lane (forward); a thread per piece column walking the slab's rows (weight gradient).
template <int CH>
__global__ void __launch_bounds__(RX_BLOCK)
gemv16_fwd_wide_kernel(const uint16_t* __restrict__ X, int64_t ldx, const float* __restrict__ w, const float* __restrict__ b,
                       float* __restrict__ Y, int64_t ldy, int64_t M, int K, int act) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float wv[CH][8];
  bool ok[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int k0 = (c * 64 + lane) * 8;
    ok[c] = k0 < K;
#pragma unroll
    for (int e = 0; e < 8; ++e) wv[c][e] = (k0 + e < K) ? round_bf16(w[k0 + e]) : 0.f;   // rows are zero padded to 8
  }
  const float bv = b ? b[0] : 0.f;
  for (int64_t r = (int64_t)blockIdx.x * (RX_BLOCK / 64) + wave; r < M; r += (int64_t)gridDim.x * (RX_BLOCK / 64)) {
    uint4 x[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) x[c] = ok[c] ? *reinterpret_cast<const uint4*>(X + r * ldx + (c * 64 + lane) * 8) : make_uint4(0u, 0u, 0u, 0u);
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const uint32_t xs[4] = {x[c].x, x[c].y, x[c].z, x[c].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) s += bf16_bits_to_f32(xs[e] & 0xffffu) * wv[c][2 * e] + bf16_bits_to_f32(xs[e] >> 16) * wv[c][2 * e + 1];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if (lane == 0) {
      float v = s + bv;
      if (act == 1) v = v > 0.f ? v : 0.f;
      Y[r * ldy] = v;
    }
  }
}

template <int CH>
__global__ void __launch_bounds__(RX_BLOCK)
gemv16_bwd_w_wide_kernel(const uint16_t* __restrict__ dY, int64_t ldd, const uint16_t* __restrict__ X, int64_t ldx, int64_t M, int K,
                         int64_t rows_per_slab, float* __restrict__ part_w, float* __restrict__ part_b) {
  const int64_t lo = (int64_t)blockIdx.x * rows_per_slab;
  const int64_t hi = (lo + rows_per_slab < M) ? lo + rows_per_slab : M;
  float s[CH][8];
  bool ok[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    ok[c] = (c * RX_BLOCK + (int)threadIdx.x) * 8 < K;
#pragma unroll
    for (int e = 0; e < 8; ++e) s[c][e] = 0.f;
  }
  float sb = 0.f;
  for (int64_t r = lo; r < hi; ++r) {
    const float gy = bf16_bits_to_f32(dY[r * ldd]);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const uint4 x = ok[c] ? *reinterpret_cast<const uint4*>(X + r * ldx + (c * RX_BLOCK + threadIdx.x) * 8) : make_uint4(0u, 0u, 0u, 0u);
      const uint32_t xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s[c][2 * e] += gy * bf16_bits_to_f32(xs[e] & 0xffffu);
        s[c][2 * e + 1] += gy * bf16_bits_to_f32(xs[e] >> 16);
      }
    }
    sb += gy;
  }
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = (c * RX_BLOCK + threadIdx.x) * 8 + e;
      if (k < K) part_w[(int64_t)blockIdx.x * K + k] = s[c][e];
    }
  if (threadIdx.x == 0 && part_b) part_b[blockIdx.x] = sb;
}

// ---- column sums of a bf16 matrix: partial[z][j]; a block covers 128 columns with 16 row lanes of 16-byte loads
template <int DUMMY>
__global__ void __launch_bounds__(RX_BLOCK)
colsum16_slab_kernel(const uint16_t* __restrict__ dY, int64_t ldd, int64_t M, int N, int64_t rows_per_slab, float* __restrict__ partial) {
  __shared__ float red[16][128 + 1];
  const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int j0 = blockIdx.x * 128 + cq * 8;
  const int64_t lo = (int64_t)blockIdx.z * rows_per_slab;
  const int64_t hi = (lo + rows_per_slab < M) ? lo + rows_per_slab : M;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (j0 < N) {   // N % 8 == 0
    int64_t r = lo + rl;
    for (; r + 48 < hi; r += 64) {
      uint4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const uint4*>(dY + (r + 16 * u) * ldd + j0);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t xs[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { s[2 * e] += bf16_bits_to_f32(xs[e] & 0xffffu); s[2 * e + 1] += bf16_bits_to_f32(xs[e] >> 16); }
      }
    }
    for (; r < hi; r += 16) {
      const uint4 v = *reinterpret_cast<const uint4*>(dY + r * ldd + j0);
      const uint32_t xs[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) { s[2 * e] += bf16_bits_to_f32(xs[e] & 0xffffu); s[2 * e + 1] += bf16_bits_to_f32(xs[e] >> 16); }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[rl][cq * 8 + e] = s[e];
  __syncthreads();
  if (threadIdx.x < 128) {
    const int j = blockIdx.x * 128 + threadIdx.x;
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q][threadIdx.x];
    if (j < N) partial[(int64_t)blockIdx.z * N + j] = t;
  }
}

// ---- wide plain products go to the library (rx_blaslt.hip); 0 done, 1 error, -1 not taken
int rx_lt_matmul(int device, hipStream_t st, int transA, int transB, int64_t m, int64_t n, int64_t k, const void* A, int64_t lda,
                 const void* B, int64_t ldb, void* D, int64_t ldd, int out_bf16, const float* bias, int relu, int beta1, void* call_ws,
                 size_t call_ws_bytes);

// [rows, n_out] = [rows, n_red] x [n_red, n_out]: where the library's kernels are the faster ones (measured at
// rows = 65536, profiles/gemm16_vs_library.py: 1024 x 432, 512 x 1024, 256 x 512 and everything wider go to the
// library; 432 x 432, 216 x 432, 108 x 216 stay here)
static const bool lt_gated_too = [] { const char* e = getenv("RX_LT_NO_GATED"); return !(e && e[0] == '1'); }();   // A/B: gated input gradients
static const bool lt_bwd_w_too = [] { const char* e = getenv("RX_LT_NO_BWD_W"); return !(e && e[0] == '1'); }();   // A/B: weight gradients
static bool lt_shape(int64_t rows, int64_t n_out, int64_t n_red) {
  const int64_t lo = n_out < n_red ? n_out : n_red, hi = n_out < n_red ? n_red : n_out;
  return rows >= 4096 && lo >= 256 && hi >= 512 && n_out % 4 == 0 && n_red % 4 == 0;
}

// dX16 = gate > 0 ? dX16 : 0 in place: the ReLU backward of the layer in front, for input gradients that came from
// the library (gemm_h16_kernel applies the gate in its epilogue)
__global__ void __launch_bounds__(RX_BLOCK)
relu_gate16_kernel(uint16_t* __restrict__ dx, int64_t ldx, const uint16_t* __restrict__ gate, int64_t ldg, int64_t rows, int c8) {
  const int64_t n = rows * c8;
  const bool small = n < (1ll << 31);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = rx_row_of(i, c8, small);
    const int c = (int)(i - r * c8) * 8;
    uint4 v = *reinterpret_cast<const uint4*>(dx + r * ldx + c);
    const uint4 g = *reinterpret_cast<const uint4*>(gate + r * ldg + c);
    auto keep = [](uint32_t x, uint32_t gg) {   // a bf16 gate passes when it is positive: not zero, sign clear
      const uint32_t glo = gg & 0xffffu, ghi = gg >> 16;
      const uint32_t mlo = (glo == 0 || (glo & 0x8000u)) ? 0u : 0xffffu, mhi = (ghi == 0 || (ghi & 0x8000u)) ? 0u : 0xffff0000u;
      return x & (mlo | mhi);
    };
    v.x = keep(v.x, g.x); v.y = keep(v.y, g.y); v.z = keep(v.z, g.z); v.w = keep(v.w, g.w);
    *reinterpret_cast<uint4*>(dx + r * ldx + c) = v;
  }
}

extern "C" int rx_linear16_fwd(const uint16_t* X16, int64_t ldx, const uint16_t* W16, const float* W, const float* b,
                               void* Y, int64_t ldy, int64_t M, int32_t N, int32_t K, int act, int out_bf16, int device,
                               void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(M >= 0 && N >= 1 && K >= 8 && K % 4 == 0 && ldx >= K && ldy >= N, "rx_linear16_fwd: bad shape M=%lld N=%d K=%d", (long long)M, N, K);
  RX_REQUIRE(act == 0 || act == 1, "rx_linear16_fwd: unknown activation %d", act);
  if (M == 0) return 0;
  RX_REQUIRE(X16 && Y && h16_ok(X16, ldx), "rx_linear16_fwd: null or unaligned activations");
  hipStream_t st = (hipStream_t)stream;
  if (gemv16_ok(N, K) && !out_bf16) {
    RX_REQUIRE(W, "rx_linear16_fwd: the one-column layer reads the fp32 weights");
    if (gemv16_wide(K)) {
      int64_t wg = rx_cdiv(M, RX_BLOCK / 64);
      if (wg > 8192) wg = 8192;
      const int ch = (int)rx_cdiv(rx_cdiv(K, 8), 64);
      if (ch <= 2) hipLaunchKernelGGL(gemv16_fwd_wide_kernel<2>, dim3((unsigned)wg), dim3(RX_BLOCK), 0, st, X16, ldx, W, b, (float*)Y, ldy, M, (int)K, act);
      else if (ch <= 4) hipLaunchKernelGGL(gemv16_fwd_wide_kernel<4>, dim3((unsigned)wg), dim3(RX_BLOCK), 0, st, X16, ldx, W, b, (float*)Y, ldy, M, (int)K, act);
      else hipLaunchKernelGGL(gemv16_fwd_wide_kernel<8>, dim3((unsigned)wg), dim3(RX_BLOCK), 0, st, X16, ldx, W, b, (float*)Y, ldy, M, (int)K, act);
      RX_LAUNCH_CHECK();
      return 0;
    }
    const int lpr = gemv16_lpr(K);
    int64_t grid = rx_cdiv(M, RX_BLOCK / lpr);
    if (grid > 4096) grid = 4096;
    RX_GEMV_DISPATCH(gemv16_fwd_kernel, lpr, dim3((unsigned)grid), dim3(RX_BLOCK), 0, st, X16, ldx, W, b, (float*)Y, ldy, M, (int)K, act);
    RX_LAUNCH_CHECK();
    return 0;
  }
  RX_REQUIRE(W16 && rx_aligned16(W16), "rx_linear16_fwd: null or unaligned bf16 weights");
  RX_REQUIRE(N % 4 == 0 && (out_bf16 ? h16_ok(Y, ldy) : f4_ok(Y, ldy)),
             "rx_linear16_fwd: N must be a multiple of 4 and output rows 16-byte addressable");
  if (lt_shape(M, N, K)) {   // Y^T (N x M, column-major) = W16^T X16^T: A = W16 as K x N, transposed; B = X16 as K x M
    const int rc = rx_lt_matmul(device, st, 1, 0, N, M, K, W16, rx_cdiv(K, 8) * 8, X16, ldx, Y, ldy, out_bf16, b, act, 0, nullptr, 0);
    if (rc >= 0) return rc;
  }
  Gemm16Args g{};
  g.A = X16; g.lda = ldx; g.B = W16; g.ldb = rx_cdiv(K, 8) * 8; g.C = Y; g.ldc = ldy; g.bias = b;
  g.M = M; g.N = N; g.K = K; g.act = act; g.k_per_slab = K;
  return out_bf16 ? gemm_h16_launch<true, true, false, false>(g, 1, st) : gemm_h16_launch<true, false, false, false>(g, 1, st);
}

extern "C" int rx_cross16_layer_fwd(const uint16_t* X16, int64_t ldx, const uint16_t* W16, const float* b, const float* x0,
                                    int64_t ld0, const float* xl, int64_t ldl, float* u, int64_t ldu, float* xn, int64_t ldn,
                                    uint16_t* xn16, int64_t ldn16, int64_t M, int32_t d, int device, void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(M >= 0 && d >= 8 && d % 8 == 0 && ldx >= d && ld0 >= d && ldl >= d && ldu >= d && ldn >= d && ldn16 >= d,
             "rx_cross16_layer_fwd: bad shape (width a multiple of 8)");
  if (M == 0) return 0;
  RX_REQUIRE(X16 && W16 && x0 && xl && u && xn && xn16, "rx_cross16_layer_fwd: null argument");
  RX_REQUIRE(h16_ok(X16, ldx) && rx_aligned16(W16) && f4_ok(x0, ld0) && f4_ok(xl, ldl) && f4_ok(u, ldu) && f4_ok(xn, ldn) &&
                 ldn16 % 4 == 0 && ((uintptr_t)xn16 % 8 == 0), "rx_cross16_layer_fwd: rows must be 16-byte addressable");
  Gemm16Args g{};
  g.A = X16; g.lda = ldx; g.B = W16; g.ldb = d; g.C = u; g.ldc = ldu; g.bias = b;
  g.M = M; g.N = d; g.K = d; g.act = 0; g.k_per_slab = d;
  g.x0 = x0; g.ld0 = ld0; g.xl = xl; g.ldl = ldl; g.xn = xn; g.ldn = ldn; g.xn16 = xn16; g.ldn16 = ldn16;
  return gemm_h16_launch<true, false, false, false, true>(g, 1, (hipStream_t)stream);
}

extern "C" int rx_linear16_bwd_x(const uint16_t* dY16, int64_t ldd, const uint16_t* W16T, const float* W, void* dX, int64_t ldx,
                                 const uint16_t* gate16, int64_t ldg, int64_t M, int32_t N, int32_t K, int accumulate,
                                 int out_bf16, int device, void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(M >= 0 && N >= 1 && K >= 1 && ldd >= N && ldx >= K, "rx_linear16_bwd_x: bad shape");
  RX_REQUIRE(!(accumulate && (out_bf16 || gate16)), "rx_linear16_bwd_x: accumulation is for ungated fp32 outputs");
  RX_REQUIRE(!gate16 || out_bf16 || N == 1, "rx_linear16_bwd_x: the gated form writes bf16 (fp32 only for the one-column layer)");
  if (M == 0) return 0;
  RX_REQUIRE(dY16 && dX, "rx_linear16_bwd_x: null argument");
  hipStream_t st = (hipStream_t)stream;
  if (gemv16_ok(N, K) && !accumulate) {
    RX_REQUIRE(W && (out_bf16 ? h16_ok(dX, ldx) : f4_ok(dX, ldx)) && (!gate16 || h16_ok(gate16, ldg)),
               "rx_linear16_bwd_x: one-column layer: fp32 weights and aligned rows");
    int64_t grid = rx_cdiv(M * ((K + 7) / 8), RX_BLOCK);
    if (grid > 8192) grid = 8192;
    if (out_bf16) hipLaunchKernelGGL(gemv16_bwd_x_kernel<false>, dim3((unsigned)grid), dim3(RX_BLOCK), 0, st, dY16, ldd, W, gate16, ldg, dX, ldx, M, (int)K);
    else hipLaunchKernelGGL(gemv16_bwd_x_kernel<true>, dim3((unsigned)grid), dim3(RX_BLOCK), 0, st, dY16, ldd, W, gate16, ldg, dX, ldx, M, (int)K);
    RX_LAUNCH_CHECK();
    return 0;
  }
  RX_REQUIRE(N % 4 == 0 && h16_ok(dY16, ldd) && W16T && rx_aligned16(W16T), "rx_linear16_bwd_x: N must be a multiple of 4, operands 16-byte aligned");
  RX_REQUIRE(K % 4 == 0 && (out_bf16 ? (h16_ok(dX, ldx) && (!gate16 || h16_ok(gate16, ldg))) : f4_ok(dX, ldx)),
             "rx_linear16_bwd_x: K must be a multiple of 4 and output (and gate) rows 16-byte addressable");
  if (lt_shape(M, K, N) && (!gate16 || (K % 8 == 0 && lt_gated_too))) {   // (the gate pass works on whole 16-byte pieces)
    // dX^T (K x M, column-major) = W^T dY^T: A = W16T as N x K, transposed; B = dY16 as N x M; the gate in a pass of its own
    const int rc = rx_lt_matmul(device, st, 1, 0, K, M, N, W16T, rx_cdiv(N, 8) * 8, dY16, ldd, dX, ldx, out_bf16, nullptr, 0, accumulate, nullptr, 0);
    if (rc > 0) return rc;
    if (rc == 0) {
      if (gate16) {
        int64_t grid = rx_cdiv(M * (K / 8), RX_BLOCK);
        if (grid > 16384) grid = 16384;
        hipLaunchKernelGGL(relu_gate16_kernel, dim3((unsigned)grid), dim3(RX_BLOCK), 0, st, (uint16_t*)dX, ldx, gate16, ldg, M, (int)(K / 8));
        RX_LAUNCH_CHECK();
      }
      return 0;
    }
  }
  Gemm16Args g{};
  g.A = dY16; g.lda = ldd; g.B = W16T; g.ldb = rx_cdiv(N, 8) * 8; g.C = dX; g.ldc = ldx; g.bias = nullptr; g.gate = gate16; g.ldg = ldg;
  g.M = M; g.N = K; g.K = N; g.act = 0; g.k_per_slab = N;
  if (gate16) return gemm_h16_launch<true, true, true, false>(g, 1, st);
  if (out_bf16) return gemm_h16_launch<true, true, false, false>(g, 1, st);
  return accumulate ? gemm_h16_launch<true, false, false, true>(g, 1, st) : gemm_h16_launch<true, false, false, false>(g, 1, st);
}

static int bwd_w16_slabs(int64_t M, int32_t N, int32_t K) {
  const int64_t tiles = rx_cdiv(N, 128) * rx_cdiv(K, 128);
  int64_t want = rx_cdiv(768, tiles);
  const int64_t max_by_rows = rx_cdiv(M, 2 * H_BK);
  if (want > max_by_rows) want = max_by_rows;
  if (want < 1) want = 1;
  if (want > 1024) want = 1024;
  return (int)want;
}

#define COLSUM16_SLABS 64
extern "C" size_t rx_linear16_bwd_w_workspace_bytes(int64_t M, int32_t N, int32_t K) {
  if (M <= 0 || N < 1 || K < 1) return 256;
  const int s = bwd_w16_slabs(M, N, K);
  const size_t bslabs = (size_t)(s > COLSUM16_SLABS ? s : COLSUM16_SLABS);
  return rx_align_up((size_t)s * N * K * sizeof(float), 256) + rx_align_up(bslabs * N * sizeof(float), 256);
}

extern "C" int rx_linear16_bwd_w(const uint16_t* dY16, int64_t ldd, const uint16_t* X16, int64_t ldx, float* dW, float* db,
                                 int64_t M, int32_t N, int32_t K, void* ws, size_t ws_bytes, int device, void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(M >= 1 && N >= 1 && K >= 8 && K % 4 == 0 && ldd >= N && ldx >= K, "rx_linear16_bwd_w: bad shape");
  RX_REQUIRE(dY16 && X16 && dW && ws && h16_ok(X16, ldx), "rx_linear16_bwd_w: null or unaligned argument");
  RX_REQUIRE(ws_bytes >= rx_linear16_bwd_w_workspace_bytes(M, N, K), "rx_linear16_bwd_w: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int s = bwd_w16_slabs(M, N, K);
  int64_t rows_per_slab = rx_cdiv(M, s);
  rows_per_slab = rx_cdiv(rows_per_slab, H_BK) * H_BK;
  const int n_slabs = (int)rx_cdiv(M, rows_per_slab);
  float* slabs = (float*)ws;
  float* bpart = (float*)((char*)ws + rx_align_up((size_t)s * N * K * sizeof(float), 256));
  const int64_t nk = (int64_t)N * K;
  if (gemv16_ok(N, K) && gemv16_wide(K)) {
    const int ch = (int)rx_cdiv(rx_cdiv(K, 8), RX_BLOCK);
    if (ch <= 1) hipLaunchKernelGGL(gemv16_bwd_w_wide_kernel<1>, dim3((unsigned)n_slabs), dim3(RX_BLOCK), 0, st, dY16, ldd, X16, ldx, M, (int)K, rows_per_slab, slabs, db ? bpart : (float*)nullptr);
    else hipLaunchKernelGGL(gemv16_bwd_w_wide_kernel<2>, dim3((unsigned)n_slabs), dim3(RX_BLOCK), 0, st, dY16, ldd, X16, ldx, M, (int)K, rows_per_slab, slabs, db ? bpart : (float*)nullptr);
    RX_LAUNCH_CHECK();
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)(rx_cdiv((int64_t)K, 64) + (db ? 1 : 0))), dim3(RX_BLOCK), 0, st, slabs,
                       (int64_t)K, n_slabs, dW, (const float*)bpart, (int64_t)(db ? 1 : 0), n_slabs, db);
    RX_LAUNCH_CHECK();
    return 0;
  }
  if (gemv16_ok(N, K)) {
    const int lpr = gemv16_lpr(K);
    RX_GEMV_DISPATCH(gemv16_bwd_w_kernel, lpr, dim3((unsigned)n_slabs), dim3(RX_BLOCK), 0, st, dY16, ldd, X16, ldx, M, (int)K, rows_per_slab,
                     slabs, db ? bpart : (float*)nullptr);
    RX_LAUNCH_CHECK();
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)(rx_cdiv((int64_t)K, 64) + (db ? 1 : 0))), dim3(RX_BLOCK), 0, st, slabs,
                       (int64_t)K, n_slabs, dW, (const float*)bpart, (int64_t)(db ? 1 : 0), n_slabs, db);
    RX_LAUNCH_CHECK();
    return 0;
  }
  RX_REQUIRE(N % 4 == 0 && h16_ok(dY16, ldd), "rx_linear16_bwd_w: N must be a multiple of 4 and dY 16-byte aligned");
  if (lt_bwd_w_too && lt_shape(M, N, K) && (int64_t)N * K >= (4 << 20)) {   // measured: the library wins at 1672 x 3344 (626 against 383 TFLOP/s), loses at 836 x 1672 and below
    // dW^T (K x N, column-major, ld K) = X16^T dY16: A = X16 as K x M; B = dY16 as N x M, transposed.  Output tiles alone
    // fill the card at these widths, so the library needs no split over the batch; the bias gradient takes the
    // column-sum pass
    // (the slab area of the caller's workspace is free on this path: lent to the library)
    const int rc = rx_lt_matmul(device, st, 0, 1, K, N, M, X16, ldx, dY16, ldd, dW, K, 0, nullptr, 0, 0, slabs,
                                rx_align_up((size_t)s * N * K * sizeof(float), 256));
    if (rc > 0) return rc;
    if (rc == 0) {
      if (db) {
        int64_t cs_rows = rx_cdiv(M, COLSUM16_SLABS);
        cs_rows = rx_cdiv(cs_rows, 16) * 16;
        const int cs = (int)rx_cdiv(M, cs_rows);
        hipLaunchKernelGGL(colsum16_slab_kernel<0>, dim3((unsigned)rx_cdiv(N, 128), 1, (unsigned)cs), dim3(RX_BLOCK), 0, st, dY16, ldd, M, (int)N, cs_rows, bpart);
        RX_LAUNCH_CHECK();
        hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)rx_cdiv(N, 64)), dim3(RX_BLOCK), 0, st, (const float*)bpart, (int64_t)N, cs, db,
                           (const float*)nullptr, (int64_t)0, 0, (float*)nullptr);
        RX_LAUNCH_CHECK();
      }
      return 0;
    }
  }
  Gemm16Args g{};
  g.A = dY16; g.lda = ldd; g.B = X16; g.ldb = ldx; g.C = slabs; g.ldc = K; g.bias = nullptr;
  g.M = N; g.N = K; g.K = M; g.act = 0; g.k_per_slab = rows_per_slab; g.c_slab_stride = nk;
  // the bias gradient's partials come out of the product's own launch: the kernel sums the columns of the dY
  // tiles it stages, one partial row per weight-gradient slab (RX_SEPARATE_COLSUM=1: the separate pass, A/B)
  static const bool separate = [] { const char* e = getenv("RX_SEPARATE_COLSUM"); return e && e[0] == '1'; }();
  const bool fused_bias = db && !separate;
  if (fused_bias) g.colsum = bpart;
  if (gemm_h16_launch<false, false, false, false>(g, n_slabs, st)) return 1;
  int cs = fused_bias ? n_slabs : 0;
  if (db && !fused_bias) {
    int64_t cs_rows = rx_cdiv(M, COLSUM16_SLABS);
    cs_rows = rx_cdiv(cs_rows, 16) * 16;
    cs = (int)rx_cdiv(M, cs_rows);
    hipLaunchKernelGGL(colsum16_slab_kernel<0>, dim3((unsigned)rx_cdiv(N, 128), 1, (unsigned)cs), dim3(RX_BLOCK), 0, st, dY16, ldd, M, (int)N, cs_rows, bpart);
    RX_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)(rx_cdiv(nk, 64) + (db ? rx_cdiv(N, 64) : 0))), dim3(RX_BLOCK), 0, st, slabs, nk,
                     n_slabs, dW, (const float*)bpart, (int64_t)(db ? N : 0), cs, db);
  RX_LAUNCH_CHECK();
  return 0;
}

// ---- DCN-v2 cross combine with the bf16 copies the matrix products read
__global__ void __launch_bounds__(RX_BLOCK)
cross16_fwd_kernel(const float* __restrict__ x0, int64_t ld0, const float* __restrict__ u, int64_t ldu,
                   const float* __restrict__ xl, int64_t ldl, float* __restrict__ xn, int64_t ldn,
                   uint16_t* __restrict__ xn16, int64_t ldn16, int64_t rows, int cq) {
  const int64_t n = rows * cq;
  const bool small = n < (1ll << 31);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = rx_row_of(i, cq, small);
    const int c = (int)(i - r * cq) * 4;
    const float4 a = *reinterpret_cast<const float4*>(x0 + r * ld0 + c);
    const float4 b = *reinterpret_cast<const float4*>(u + r * ldu + c);
    const float4 d = *reinterpret_cast<const float4*>(xl + r * ldl + c);
    const float4 o = make_float4(a.x * b.x + d.x, a.y * b.y + d.y, a.z * b.z + d.z, a.w * b.w + d.w);
    *reinterpret_cast<float4*>(xn + r * ldn + c) = o;
    *reinterpret_cast<uint2*>(xn16 + r * ldn16 + c) = make_uint2((uint32_t)f32_to_bf16_rne(o.x) | ((uint32_t)f32_to_bf16_rne(o.y) << 16),
                                                                 (uint32_t)f32_to_bf16_rne(o.z) | ((uint32_t)f32_to_bf16_rne(o.w) << 16));
  }
}

__global__ void __launch_bounds__(RX_BLOCK)
cross16_bwd_kernel(const float* __restrict__ dxn, int64_t ldd, const float* __restrict__ x0, int64_t ld0,
                   const float* __restrict__ u, int64_t ldu, uint16_t* __restrict__ du16, int64_t lddu,
                   float* __restrict__ dx0_acc, int64_t ldx0, float* __restrict__ dxl, int64_t ldxl,
                   int64_t rows, int cq, int accumulate, int identity) {
  const int64_t n = rows * cq;
  const bool small = n < (1ll << 31);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = rx_row_of(i, cq, small);
    const int c = (int)(i - r * cq) * 4;
    const float4 d = *reinterpret_cast<const float4*>(dxn + r * ldd + c);
    const float4 a = *reinterpret_cast<const float4*>(x0 + r * ld0 + c);
    const float4 b = *reinterpret_cast<const float4*>(u + r * ldu + c);
    *reinterpret_cast<uint2*>(du16 + r * lddu + c) = make_uint2((uint32_t)f32_to_bf16_rne(d.x * a.x) | ((uint32_t)f32_to_bf16_rne(d.y * a.y) << 16),
                                                                (uint32_t)f32_to_bf16_rne(d.z * a.z) | ((uint32_t)f32_to_bf16_rne(d.w * a.w) << 16));
    float4 g0 = make_float4(d.x * b.x, d.y * b.y, d.z * b.z, d.w * b.w);
    if (identity) { g0.x += d.x; g0.y += d.y; g0.z += d.z; g0.w += d.w; }
    float4* acc = reinterpret_cast<float4*>(dx0_acc + r * ldx0 + c);
    if (accumulate) { const float4 o = *acc; g0.x += o.x; g0.y += o.y; g0.z += o.z; g0.w += o.w; }
    *acc = g0;
    if (dxl) *reinterpret_cast<float4*>(dxl + r * ldxl + c) = d;
  }
}

extern "C" int rx_cross16_fwd(const float* x0, int64_t ld0, const float* u, int64_t ldu, const float* xl, int64_t ldl,
                              float* xn, int64_t ldn, uint16_t* xn16, int64_t ldn16, int64_t rows, int32_t cols, int device,
                              void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(rows >= 0 && cols >= 4 && cols % 4 == 0 && ld0 >= cols && ldu >= cols && ldl >= cols && ldn >= cols && ldn16 >= cols,
             "rx_cross16_fwd: bad shape (cols a multiple of 4)");
  if (rows == 0) return 0;
  RX_REQUIRE(x0 && u && xl && xn && xn16, "rx_cross16_fwd: null argument");
  RX_REQUIRE(f4_ok(x0, ld0) && f4_ok(u, ldu) && f4_ok(xl, ldl) && f4_ok(xn, ldn) && ldn16 % 4 == 0 && ((uintptr_t)xn16 % 8 == 0),
             "rx_cross16_fwd: rows must be 16-byte addressable");
  int64_t grid = rx_cdiv(rows * (cols / 4), RX_BLOCK);
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(cross16_fwd_kernel, dim3((unsigned)grid), dim3(RX_BLOCK), 0, (hipStream_t)stream, x0, ld0, u, ldu, xl, ldl, xn, ldn,
                     xn16, ldn16, rows, (int)(cols / 4));
  RX_LAUNCH_CHECK();
  return 0;
}

extern "C" int rx_cross16_bwd(const float* dxn, int64_t ldd, const float* x0, int64_t ld0, const float* u, int64_t ldu,
                              uint16_t* du16, int64_t lddu, float* dx0_acc, int64_t ldx0, float* dxl, int64_t ldxl,
                              int64_t rows, int32_t cols, int accumulate, int identity, int device, void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(rows >= 0 && cols >= 4 && cols % 4 == 0 && ldd >= cols && ld0 >= cols && ldu >= cols && lddu >= cols && ldx0 >= cols &&
                 (!dxl || ldxl >= cols), "rx_cross16_bwd: bad shape (cols a multiple of 4)");
  if (rows == 0) return 0;
  RX_REQUIRE(dxn && x0 && u && du16 && dx0_acc, "rx_cross16_bwd: null argument");
  RX_REQUIRE(f4_ok(dxn, ldd) && f4_ok(x0, ld0) && f4_ok(u, ldu) && f4_ok(dx0_acc, ldx0) && (!dxl || f4_ok(dxl, ldxl)) && lddu % 4 == 0 &&
                 ((uintptr_t)du16 % 8 == 0), "rx_cross16_bwd: rows must be 16-byte addressable");
  int64_t grid = rx_cdiv(rows * (cols / 4), RX_BLOCK);
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(cross16_bwd_kernel, dim3((unsigned)grid), dim3(RX_BLOCK), 0, (hipStream_t)stream, dxn, ldd, x0, ld0, u, ldu, du16, lddu,
                     dx0_acc, ldx0, dxl, ldxl, rows, (int)(cols / 4), accumulate, identity);
  RX_LAUNCH_CHECK();
  return 0;
}
