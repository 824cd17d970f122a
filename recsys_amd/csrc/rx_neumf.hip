// NeuMF train step as ONE kernel (gfx950): gathers, GMF product, both towers, the output layer, the BCE
// loss, and the whole backward pass down to per-row embedding gradients and weight-gradient partials.
//
// The list-wise NeuMF of the reference (recsys/models/neuralcf/neumf.py:68-106) is, per (user, item) pair,
//   x   = [user_mlp[u] | item_mlp[i]]                       2D
//   h1  = dropout(relu(W1 x + b1))                          D        linear_mlp.0 / ReLU / Dropout
//   h2  = W2 h1 + b2                                        D/2      linear_mlp.3 (no activation behind it)
//   gm  = Wg (user_gmf[u] * item_gmf[i]) + bg               D/2      linear_gmf
//   logit = wf . [gm | h2] + bf                                      linear_final on cat([gmf, mlp])
// and its loss the mean BCE over all B x N pairs (neumf.py:75-79), so dL/dlogit = (sigmoid(logit) - y) / (B N)
// is known as soon as the pair's logit is: nothing in the backward pass of one pair waits for another pair.
// The only sums across pairs are the weight gradients, the gradient of a user's rows over that user's N items,
// and the loss.  A workgroup therefore takes a tile of whole samples (64 / N samples, at most 64 pairs), keeps
// every activation of the tile in LDS, runs all nine matrix products of forward and backward on
// v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulation), keeps its share of the weight gradients in
// accumulator registers across its tiles, and leaves
//   du_g, du_m [B, D]      gradient of the user's GMF / MLP row, summed over the sample's N pairs (pair order)
//   di_g, di_m [B N, D]    gradient of every pair's item rows
//   wpart [grid][WP]       per-workgroup weight-gradient partials, added in workgroup order by the finish kernel
//   loss partials          fp64, one per workgroup
// The embedding gradients then take the usual deterministic path (sort + segmented reduce, fused row update).
// Written for D = 64 (tables of 64 floats, tower 128 -> 64 -> 32, GMF 64 -> 32): 151 KB of LDS per workgroup.
//
// LDS operands are read in the two patterns an MFMA lane needs -- (index = lane % 32, k = lane / 32) and
// (k = lane / 32, index = lane % 32) -- from the same arrays; every array has an ODD row pitch, so both
// patterns spread a wavefront's reads over all banks and each weight matrix is kept once (W1 as [in][out]).
#include "rx_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define NM_D 64
#define NM_P 64       // pair rows of a tile (two MFMA row tiles)
#define NM_LDX 129    // pitch of the tower input / its gradient [pair][2 D]
#define NM_LD 65      // pitch of the 64-wide activation arrays
#define NM_LDW 33     // pitch of the [in = 64][out = 32] weight arrays
#define NM_WP (64 * 128 + 64 + 32 * 64 + 32 + 32 * 64 + 32 + 64 + 1)   // floats of one weight-gradient partial
#define NM_MAX_GRID 256

// LDS map (floats)
#define NM_OFF_X 0
#define NM_OFF_ZG (NM_OFF_X + NM_P * NM_LDX)
#define NM_OFF_H1 (NM_OFF_ZG + NM_P * NM_LD)
#define NM_OFF_HG (NM_OFF_H1 + NM_P * NM_LD)
#define NM_OFF_DHG (NM_OFF_HG + NM_P * NM_LD)
#define NM_OFF_W1T (NM_OFF_DHG + NM_P * NM_LD)
#define NM_OFF_W2T (NM_OFF_W1T + 128 * NM_LD)
#define NM_OFF_WGT (NM_OFF_W2T + 64 * NM_LDW)
#define NM_OFF_B1 (NM_OFF_WGT + 64 * NM_LDW)
#define NM_OFF_B2 (NM_OFF_B1 + 64)
#define NM_OFF_BG (NM_OFF_B2 + 32)
#define NM_OFF_WF (NM_OFF_BG + 32)
#define NM_OFF_DL (NM_OFF_WF + 64)
#define NM_LDS_FLOATS (NM_OFF_DL + 64)

struct NmArgs {
  const float *Ug, *Ig, *Um, *Im;
  int64_t n_users, n_items;
  const int64_t* users;   // [B]
  const int64_t* items;   // [B * N]
  const void* labels;     // [B * N] f64 or f32
  int64_t B;
  int N;
  const float *W1, *b1, *W2, *b2, *Wg, *bg, *wf, *bf;
  float p_drop;
  const int64_t* seed;    // device counter: the dropout stream of this call (NULL: 0)
  float *du_g, *du_m, *di_g, *di_m;
  float* wpart;
  double* loss_part;
  float* logits;          // optional [B * N]
  int32_t* err_flag;
};

// splitmix64 finaliser over (seed, counter): one 64-bit draw per (pair, hidden unit)
__device__ __forceinline__ uint64_t nm_mix(uint64_t seed, uint64_t counter) {
  uint64_t z = seed * 0xD1342543DE82EF95ull + (counter + 1ull) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

#define NM_ROW(r, hh) (((r) & 3) + 8 * ((r) >> 2) + 4 * (hh))   // row of accumulator register r inside a 32 x 32 tile

template <bool F64>
__global__ void __launch_bounds__(RX_BLOCK)
neumf_step_kernel(NmArgs a) {
  extern __shared__ float lds[];
  float* X = lds + NM_OFF_X;
  float* ZG = lds + NM_OFF_ZG;
  float* H1 = lds + NM_OFF_H1;
  float* HG = lds + NM_OFF_HG;
  float* DHG = lds + NM_OFF_DHG;
  float* W1T = lds + NM_OFF_W1T;
  float* W2T = lds + NM_OFF_W2T;
  float* WGT = lds + NM_OFF_WGT;
  float* B1 = lds + NM_OFF_B1;
  float* B2 = lds + NM_OFF_B2;
  float* BG = lds + NM_OFF_BG;
  float* WF = lds + NM_OFF_WF;
  float* DL = lds + NM_OFF_DL;
  __shared__ double lsum_w[RX_BLOCK / 64];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5, l31 = lane & 31;
  const int N = a.N;
  const int S = NM_P / N;                       // whole samples per tile
  const int64_t tiles = (a.B + S - 1) / S;
  const double inv_m = 1.0 / ((double)a.B * (double)N);

  // weights: W1 [64][128] -> W1T[c][h]; W2, Wg [32][64] -> [c][o]
  for (int i = tid; i < 64 * 128; i += RX_BLOCK) W1T[(i & 127) * NM_LD + (i >> 7)] = a.W1[i];
  for (int i = tid; i < 32 * 64; i += RX_BLOCK) {
    W2T[(i & 63) * NM_LDW + (i >> 6)] = a.W2[i];
    WGT[(i & 63) * NM_LDW + (i >> 6)] = a.Wg[i];
  }
  if (tid < 64) { B1[tid] = a.b1[tid]; WF[tid] = a.wf[tid]; }
  if (tid < 32) { B2[tid] = a.b2[tid]; BG[tid] = a.bg[tid]; }
  const float bf = a.bf[0];
  const uint64_t seed = a.seed ? (uint64_t)a.seed[0] : 0ull;
  const bool drop = a.p_drop > 0.f;
  const uint32_t thr = drop ? (uint32_t)((double)a.p_drop * 4294967296.0) : 0u;
  const float inv_keep = drop ? 1.f / (1.f - a.p_drop) : 1.f;

  f32x16 dW1a[2], dWs;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dW1a[0][r] = 0.f; dW1a[1][r] = 0.f; dWs[r] = 0.f; }
  float bsum = 0.f;       // this thread's bias-type gradient: dwf[tid] | dbf | db2 | dbg | db1 by thread range
  double lsum = 0.0;
  bool bad = false;
  __syncthreads();

  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t s0 = tile * S;
    const int ns = (int)((a.B - s0 < S) ? a.B - s0 : S);
    const int PT = ns * N;
    const int64_t p0 = s0 * N;                  // first pair of the tile (pairs of a tile are contiguous)
    // ---- 1. gather: tower input and GMF product
    for (int q = tid; q < NM_P * 16; q += RX_BLOCK) {
      const int p = q >> 4, c = (q & 15) * 4;
      float4 um = make_float4(0.f, 0.f, 0.f, 0.f), im = um, ug = um, ig = um;
      if (p < PT) {
        const int64_t u = a.users[s0 + p / N], it = a.items[p0 + p];
        const bool ok = u >= 0 && u < a.n_users && it >= 0 && it < a.n_items;
        if (ok) {
          um = *reinterpret_cast<const float4*>(a.Um + u * NM_D + c);
          im = *reinterpret_cast<const float4*>(a.Im + it * NM_D + c);
          ug = *reinterpret_cast<const float4*>(a.Ug + u * NM_D + c);
          ig = *reinterpret_cast<const float4*>(a.Ig + it * NM_D + c);
        } else {
          bad = true;
        }
      }
      float* xr = X + p * NM_LDX + c;
      xr[0] = um.x; xr[1] = um.y; xr[2] = um.z; xr[3] = um.w;
      xr[64] = im.x; xr[65] = im.y; xr[66] = im.z; xr[67] = im.w;
      float* zr = ZG + p * NM_LD + c;
      zr[0] = ug.x * ig.x; zr[1] = ug.y * ig.y; zr[2] = ug.z * ig.z; zr[3] = ug.w * ig.w;
    }
    __syncthreads();
    // ---- 2. h1 = dropout(relu(x W1^T + b1)): wave -> 32 x 32 tile (rt, ct) of [64 pairs, 64 units]
    {
      const int rt = wave >> 1, ct = wave & 1;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const float* ap = X + (rt * 32 + l31) * NM_LDX + hh;
      const float* bp = W1T + hh * NM_LD + ct * 32 + l31;
#pragma unroll 8
      for (int kk = 0; kk < 128; kk += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[kk], bp[kk * NM_LD], acc, 0, 0, 0);
      const int col = ct * 32 + l31;
      const float bv = B1[col];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rt * 32 + NM_ROW(r, hh);
        float v = acc[r] + bv;
        v = v > 0.f ? v : 0.f;
        if (drop) {
          const uint32_t draw = (uint32_t)(nm_mix(seed, (uint64_t)(p0 + row) * 64ull + (uint64_t)col) >> 32);
          v = draw >= thr ? v * inv_keep : 0.f;
        }
        H1[row * NM_LD + col] = row < PT ? v : 0.f;
      }
    }
    __syncthreads();
    // ---- 3. h2 = h1 W2^T + b2 (waves 0, 1) and gm = zg Wg^T + bg (waves 2, 3): 32 rows each, 32 units
    {
      const int rt = wave & 1;
      const bool mlp = wave < 2;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const float* ap = (mlp ? H1 : ZG) + (rt * 32 + l31) * NM_LD + hh;
      const float* bp = (mlp ? W2T : WGT) + hh * NM_LDW + l31;
#pragma unroll 8
      for (int kk = 0; kk < 64; kk += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[kk], bp[kk * NM_LDW], acc, 0, 0, 0);
      const float bv = mlp ? B2[l31] : BG[l31];
      const int col = (mlp ? 32 : 0) + l31;         // linear_final reads [gmf | mlp]
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rt * 32 + NM_ROW(r, hh);
        HG[row * NM_LD + col] = row < PT ? acc[r] + bv : 0.f;
      }
    }
    __syncthreads();
    // ---- 4. logit, BCE term (fp64), dL/dlogit: one thread per pair
    if (tid < NM_P) {
      float dl = 0.f;
      if (tid < PT) {
        float x = bf;
        const float* hp = HG + tid * NM_LD;
#pragma unroll 8
        for (int j = 0; j < 64; ++j) x += hp[j] * WF[j];
        if (a.logits) a.logits[p0 + tid] = x;
        const double xd = (double)x;
        const double y = F64 ? reinterpret_cast<const double*>(a.labels)[p0 + tid] : (double)reinterpret_cast<const float*>(a.labels)[p0 + tid];
        const double e = exp(-fabs(xd));
        lsum += fmax(xd, 0.0) - xd * y + log1p(e);
        const double sig = xd >= 0.0 ? 1.0 / (1.0 + e) : e / (1.0 + e);
        dl = (float)((sig - y) * inv_m);
      }
      DL[tid] = dl;
    }
    __syncthreads();
    // ---- 5. d[gm | h2] = dl * wf; gradients of wf and bf
    for (int q = tid; q < NM_P * 64; q += RX_BLOCK) DHG[(q >> 6) * NM_LD + (q & 63)] = DL[q >> 6] * WF[q & 63];
    if (tid < 64) {
      float t = 0.f;
      for (int p = 0; p < NM_P; ++p) t += DL[p] * HG[p * NM_LD + tid];
      bsum += t;
    } else if (tid == 64) {
      float t = 0.f;
      for (int p = 0; p < NM_P; ++p) t += DL[p];
      bsum += t;
    }
    __syncthreads();
    // ---- 6. dW2 += dh2^T h1, dWg += dgm^T zg (one 32 x 32 tile per wave, k = pairs); dh1 = dh2 W2, dzg = dgm Wg
    f32x16 acc1, acc2;
    {
      const bool mlp = wave < 2;
      const int ct = wave & 1;
      const float* ap = DHG + hh * NM_LD + (mlp ? 32 : 0) + l31;         // A[i = unit][k = pair]
      const float* bp = (mlp ? H1 : ZG) + hh * NM_LD + ct * 32 + l31;    // B[k = pair][j = input]
#pragma unroll 8
      for (int kk = 0; kk < NM_P; kk += 2) dWs = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[kk * NM_LD], bp[kk * NM_LD], dWs, 0, 0, 0);
      const int rt = wave >> 1, c2 = wave & 1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc1[r] = 0.f; acc2[r] = 0.f; }
      const float* a1 = DHG + (rt * 32 + l31) * NM_LD + 32 + hh;         // A[i = pair][k = unit of h2]
      const float* b1p = W2T + (c2 * 32 + l31) * NM_LDW + hh;            // B[k = unit][j = input] = W2[unit][input]
      const float* a2 = DHG + (rt * 32 + l31) * NM_LD + hh;
      const float* b2p = WGT + (c2 * 32 + l31) * NM_LDW + hh;
#pragma unroll 8
      for (int kk = 0; kk < 32; kk += 2) {
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[kk], b1p[kk], acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[kk], b2p[kk], acc2, 0, 0, 0);
      }
      if (tid >= 65 && tid < 129) {      // db2[o] (65..96), dbg[o] (97..128): column sums of d[gm | h2]
        const int col = tid < 97 ? 32 + (tid - 65) : tid - 97;
        float t = 0.f;
        for (int p = 0; p < NM_P; ++p) t += DHG[p * NM_LD + col];
        bsum += t;
      }
    }
    __syncthreads();      // every read of h1 and zg is done: they now take the gradients
    {
      const int rt = wave >> 1, c2 = wave & 1;
      const int col = c2 * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rt * 32 + NM_ROW(r, hh);
        const float h = H1[row * NM_LD + col];
        H1[row * NM_LD + col] = h > 0.f ? acc1[r] * inv_keep : 0.f;     // through the dropout scale and the ReLU
        ZG[row * NM_LD + col] = acc2[r];
      }
    }
    __syncthreads();
    // ---- 7. dW1 += dz1^T x (two tiles per wave); dx = dz1 W1 (two tiles per wave); db1
    {
      const int ht = wave & 1, cp = wave >> 1;
      const float* ap = H1 + hh * NM_LD + ht * 32 + l31;                 // A[i = unit][k = pair]
      const float* bp = X + hh * NM_LDX + cp * 64 + l31;                 // B[k = pair][j = input]
#pragma unroll 4
      for (int kk = 0; kk < NM_P; kk += 2) {
        const float av = ap[kk * NM_LD];
        dW1a[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bp[kk * NM_LDX], dW1a[0], 0, 0, 0);
        dW1a[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bp[kk * NM_LDX + 32], dW1a[1], 0, 0, 0);
      }
      const int rt = wave & 1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc1[r] = 0.f; acc2[r] = 0.f; }
      const float* a1 = H1 + (rt * 32 + l31) * NM_LD + hh;               // A[i = pair][k = unit]
      const float* b1p = W1T + (cp * 64 + l31) * NM_LD + hh;             // B[k = unit][j = input] = W1[unit][input]
#pragma unroll 4
      for (int kk = 0; kk < 64; kk += 2) {
        const float av = a1[kk];
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1p[kk], acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1p[32 * NM_LD + kk], acc2, 0, 0, 0);
      }
      if (tid >= 129 && tid < 193) {     // db1[h]
        float t = 0.f;
        for (int p = 0; p < NM_P; ++p) t += H1[p * NM_LD + (tid - 129)];
        bsum += t;
      }
      __syncthreads();    // every read of x is done: it now takes dx
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rt * 32 + NM_ROW(r, hh);
        X[row * NM_LDX + cp * 64 + l31] = acc1[r];
        X[row * NM_LDX + cp * 64 + 32 + l31] = acc2[r];
      }
    }
    __syncthreads();
    // ---- 8. embedding gradients out: items per pair, users summed over the sample's pairs in pair order
    for (int q = tid; q < PT * 16; q += RX_BLOCK) {
      const int p = q >> 4, c = (q & 15) * 4;
      const int64_t u = a.users[s0 + p / N];
      const bool ok = u >= 0 && u < a.n_users;
      const float4 ug = ok ? *reinterpret_cast<const float4*>(a.Ug + u * NM_D + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float* xr = X + p * NM_LDX + 64 + c;
      const float* zr = ZG + p * NM_LD + c;
      *reinterpret_cast<float4*>(a.di_m + (p0 + p) * NM_D + c) = make_float4(xr[0], xr[1], xr[2], xr[3]);
      *reinterpret_cast<float4*>(a.di_g + (p0 + p) * NM_D + c) = make_float4(zr[0] * ug.x, zr[1] * ug.y, zr[2] * ug.z, zr[3] * ug.w);
    }
    for (int q = tid; q < ns * 16; q += RX_BLOCK) {
      const int s = q >> 4, c = (q & 15) * 4;
      float4 m = make_float4(0.f, 0.f, 0.f, 0.f), g = m;
      for (int n = 0; n < N; ++n) {
        const int p = s * N + n;
        const int64_t it = a.items[p0 + p];
        const bool ok = it >= 0 && it < a.n_items;
        const float4 ig = ok ? *reinterpret_cast<const float4*>(a.Ig + it * NM_D + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float* xr = X + p * NM_LDX + c;
        const float* zr = ZG + p * NM_LD + c;
        m.x += xr[0]; m.y += xr[1]; m.z += xr[2]; m.w += xr[3];
        g.x += zr[0] * ig.x; g.y += zr[1] * ig.y; g.z += zr[2] * ig.z; g.w += zr[3] * ig.w;
      }
      *reinterpret_cast<float4*>(a.du_m + (s0 + s) * NM_D + c) = m;
      *reinterpret_cast<float4*>(a.du_g + (s0 + s) * NM_D + c) = g;
    }
    __syncthreads();
  }

  if (bad && a.err_flag) atomicOr(a.err_flag, RX_ERR_INDEX);
  // ---- this workgroup's weight-gradient partial: [W1 | b1 | W2 | b2 | Wg | bg | wf | bf]
  float* wp = a.wpart + (size_t)blockIdx.x * NM_WP;
  {
    const int ht = wave & 1, cp = wave >> 1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int h = ht * 32 + NM_ROW(r, hh);
      wp[h * 128 + cp * 64 + l31] = dW1a[0][r];
      wp[h * 128 + cp * 64 + 32 + l31] = dW1a[1][r];
    }
    float* w2 = wp + 64 * 128 + 64 + (wave < 2 ? 0 : 32 * 64 + 32);
    const int ct = wave & 1;
#pragma unroll
    for (int r = 0; r < 16; ++r) w2[NM_ROW(r, hh) * 64 + ct * 32 + l31] = dWs[r];
  }
  float* pb1 = wp + 64 * 128;
  float* pb2 = pb1 + 64 + 32 * 64;
  float* pbg = pb2 + 32 + 32 * 64;
  float* pwf = pbg + 32;
  if (tid < 64) pwf[tid] = bsum;
  else if (tid == 64) pwf[64] = bsum;
  else if (tid < 97) pb2[tid - 65] = bsum;
  else if (tid < 129) pbg[tid - 97] = bsum;
  else if (tid < 193) pb1[tid - 129] = bsum;
  lsum = rx_wave_sum_f64(lsum);
  if (lane == 0) lsum_w[wave] = lsum;
  __syncthreads();
  if (tid == 0) a.loss_part[blockIdx.x] = (lsum_w[0] + lsum_w[1]) + (lsum_w[2] + lsum_w[3]);
}

// Adds the workgroups' partials in workgroup order (fixed tree: bitwise reproducible), finishes the loss and
// advances the dropout counter.
__global__ void __launch_bounds__(RX_BLOCK)
neumf_finish_kernel(const float* __restrict__ wpart, int n_parts, float* __restrict__ wgrad, const double* __restrict__ loss_part,
                    double count, double* __restrict__ loss, int64_t* seed) {
  const int i = blockIdx.x * RX_BLOCK + threadIdx.x;
  if (i < NM_WP) {
    float s = 0.f;
    for (int b = 0; b < n_parts; ++b) s += wpart[(size_t)b * NM_WP + i];
    wgrad[i] = s;
  }
  if (blockIdx.x == 0 && threadIdx.x < 64) {
    double s = 0.0;
    for (int b = threadIdx.x; b < n_parts; b += 64) s += loss_part[b];
    s = rx_wave_sum_f64(s);
    if (threadIdx.x == 0) {
      loss[0] = s / count;
      if (seed) seed[0] = seed[0] + 1;
    }
  }
}

static int nm_grid(int64_t B, int N) {
  const int S = NM_P / N;
  const int64_t tiles = (B + S - 1) / S;
  return (int)(tiles < NM_MAX_GRID ? tiles : NM_MAX_GRID);
}

extern "C" int rx_neumf_step_supported(int32_t dim, int32_t N) { return dim == NM_D && N >= 1 && N <= NM_P; }

extern "C" int64_t rx_neumf_step_weight_floats(void) { return NM_WP; }

extern "C" size_t rx_neumf_step_workspace_bytes(int64_t B, int32_t N) {
  if (B < 1 || N < 1 || N > NM_P) return 256;
  const int g = nm_grid(B, N);
  return rx_align_up((size_t)g * NM_WP * sizeof(float), 256) + rx_align_up((size_t)g * sizeof(double), 256);
}

extern "C" int rx_neumf_step(const rx_neumf_args* r, void* ws, size_t ws_bytes, int32_t* err_flag, int device, void* stream) {
  RX_ENTER(device);
  RX_REQUIRE(r, "rx_neumf_step: null args");
  RX_REQUIRE(r->dim == NM_D, "rx_neumf_step: written for embedding_dim %d (tower 2D -> D -> D/2), got %d", NM_D, r->dim);
  RX_REQUIRE(r->B >= 1 && r->N >= 1 && r->N <= NM_P, "rx_neumf_step: B >= 1 and 1 <= N <= %d", NM_P);
  RX_REQUIRE(r->B * (int64_t)r->N < (1ll << 31), "rx_neumf_step: too many pairs");
  RX_REQUIRE(r->user_gmf && r->item_gmf && r->user_mlp && r->item_mlp && r->users && r->items && r->labels, "rx_neumf_step: null table, index or label pointer");
  RX_REQUIRE(r->w1 && r->b1 && r->w2 && r->b2 && r->wg && r->bg && r->wf && r->bf, "rx_neumf_step: null weight pointer");
  RX_REQUIRE(r->du_gmf && r->du_mlp && r->di_gmf && r->di_mlp && r->wgrad && r->loss && ws, "rx_neumf_step: null output pointer");
  RX_REQUIRE(rx_aligned16(r->user_gmf) && rx_aligned16(r->item_gmf) && rx_aligned16(r->user_mlp) && rx_aligned16(r->item_mlp) &&
             rx_aligned16(r->du_gmf) && rx_aligned16(r->du_mlp) && rx_aligned16(r->di_gmf) && rx_aligned16(r->di_mlp),
             "rx_neumf_step: tables and gradient buffers must be 16-byte aligned");
  RX_REQUIRE(r->n_users >= 1 && r->n_items >= 1, "rx_neumf_step: empty table");
  RX_REQUIRE(r->p_drop >= 0.f && r->p_drop < 1.f, "rx_neumf_step: dropout probability outside [0, 1)");
  RX_REQUIRE(ws_bytes >= rx_neumf_step_workspace_bytes(r->B, r->N) && ((uintptr_t)ws & 255) == 0, "rx_neumf_step: workspace too small or not 256-byte aligned");
  const int grid = nm_grid(r->B, r->N);
  NmArgs a{};
  a.Ug = r->user_gmf; a.Ig = r->item_gmf; a.Um = r->user_mlp; a.Im = r->item_mlp;
  a.n_users = r->n_users; a.n_items = r->n_items; a.users = r->users; a.items = r->items; a.labels = r->labels;
  a.B = r->B; a.N = r->N;
  a.W1 = r->w1; a.b1 = r->b1; a.W2 = r->w2; a.b2 = r->b2; a.Wg = r->wg; a.bg = r->bg; a.wf = r->wf; a.bf = r->bf;
  a.p_drop = r->p_drop; a.seed = r->seed;
  a.du_g = r->du_gmf; a.du_m = r->du_mlp; a.di_g = r->di_gmf; a.di_m = r->di_mlp;
  a.wpart = (float*)ws;
  a.loss_part = (double*)((char*)ws + rx_align_up((size_t)grid * NM_WP * sizeof(float), 256));
  a.logits = r->logits; a.err_flag = err_flag;
  const size_t lds = NM_LDS_FLOATS * sizeof(float);
  static bool attr_set[64] = {};
  if (device >= 64 || !attr_set[device]) {   // LDS beyond 64 KiB per workgroup has to be asked for, per device
    RX_HIP(hipFuncSetAttribute((const void*)neumf_step_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    RX_HIP(hipFuncSetAttribute((const void*)neumf_step_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (device < 64) attr_set[device] = true;
  }
  hipStream_t st = (hipStream_t)stream;
  if (r->labels_f64) hipLaunchKernelGGL(neumf_step_kernel<true>, dim3(grid), dim3(RX_BLOCK), lds, st, a);
  else hipLaunchKernelGGL(neumf_step_kernel<false>, dim3(grid), dim3(RX_BLOCK), lds, st, a);
  RX_LAUNCH_CHECK();
  hipLaunchKernelGGL(neumf_finish_kernel, dim3((unsigned)rx_cdiv(NM_WP, RX_BLOCK)), dim3(RX_BLOCK), 0, st, (const float*)a.wpart, grid,
                     r->wgrad, (const double*)a.loss_part, (double)r->B * (double)r->N, r->loss, const_cast<int64_t*>(r->seed));
  RX_LAUNCH_CHECK();
  return 0;
}
