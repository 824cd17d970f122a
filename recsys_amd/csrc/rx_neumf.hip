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
//
// Nothing non-linear stands between linear_mlp.3 / linear_gmf and linear_final (neumf.py:44 removes the last
// ReLU and Dropout), so with  t2 = W2^T wf[D/2:],  tg = Wg^T wf[:D/2],  c0 = bf + wf[:D/2].bg + wf[D/2:].b2
//   logit = c0 + tg . zg + t2 . h1           zg = user_gmf[u] * item_gmf[i]
//   dh1 = dl t2,   dzg = dl tg               dl = dL/dlogit of the pair
//   dW2 = wf[D/2:] (x) v,   db2 = wf[D/2:] sum(dl),   v  = sum_pairs dl h1
//   dWg = wf[:D/2] (x) vg,  dbg = wf[:D/2] sum(dl),   vg = sum_pairs dl zg
//   dwf = [Wg vg + bg sum(dl) | W2 v + b2 sum(dl)],   dbf = sum(dl)
// which leaves three matrix products per pair tile -- the first tower layer forward, and its two gradients
// (dW1 += dz1^T x, dx = dz1 W1) -- on v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulation).  The
// only sums across pairs are those weight-gradient terms, the gradient of a user's rows over that user's N
// items, and the loss.  A workgroup takes tiles of whole samples (64 / N samples, at most 64 pairs), keeps the
// tile's activations in LDS, keeps dW1 in accumulator registers across its tiles, fetches the rows of its next
// tile while the gradient products of the current one run, and leaves
//   du_g, du_m [B, D]      gradient of the user's GMF / MLP row, summed over the sample's N pairs (pair order)
//   di_g, di_m [B N, D]    gradient of every pair's item rows
//   raw [grid][NM_RAW]     per-workgroup [dW1 | db1 | v | vg | sum(dl)], added in workgroup order by the finish
//                          kernel, which also expands v, vg, sum(dl) into the gradients of W2, b2, Wg, bg, wf, bf
//   loss partials          fp64, one per workgroup
// The embedding gradients then take the usual deterministic path (sort + segmented reduce, fused row update).
// Written for D = 64 (tables of 64 floats, tower 128 -> 64 -> 32, GMF 64 -> 32): 153 KB of LDS per workgroup.
//
// LDS operands are read in the two patterns an MFMA lane needs -- (index = lane % 32, k = lane / 32) and
// (k = lane / 32, index = lane % 32) -- from the same arrays; every array has an ODD row pitch, so both
// patterns spread a wavefront's reads over all banks and W1 is kept once (as [in][out]).
#include "rx_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

#define NM_D 64
#define NM_P 64       // pair rows of a tile (two MFMA row tiles)
#define NM_LDX 129    // pitch of the tower input / its gradient [pair][2 D]
#define NM_LD 65      // pitch of the 64-wide arrays
#define NM_LG 72      // pitch (bf16 elements) of the two gate arrays: 16-byte reads of 32 consecutive rows touch every bank once
#define NM_WP (64 * 128 + 64 + 32 * 64 + 32 + 32 * 64 + 32 + 64 + 1)   // floats of the weight gradient [W1|b1|W2|b2|Wg|bg|wf|bf]
#define NM_RAW (64 * 128 + 64 + 64 + 64 + 1)                            // floats of one workgroup's partial [dW1|db1|v|vg|sum dl]
#define NM_RAW_V (64 * 128 + 64)
#define NM_RAW_VG (NM_RAW_V + 64)
#define NM_RAW_SDL (NM_RAW_VG + 64)
#define NM_MAX_GRID 256

// LDS map (floats)
#define NM_OFF_X 0
#define NM_OFF_H1 (NM_OFF_X + NM_P * NM_LDX)
#define NM_OFF_GU (NM_OFF_H1 + NM_P * NM_LD)
#define NM_OFF_W1T (NM_OFF_GU + NM_P * NM_LD)
#define NM_OFF_B1 (NM_OFF_W1T + 128 * NM_LD)
#define NM_OFF_T2 (NM_OFF_B1 + 64)
#define NM_OFF_TG (NM_OFF_T2 + 64)
#define NM_OFF_DL (NM_OFF_TG + 64)
#define NM_OFF_ZD (NM_OFF_DL + 64)
#define NM_OFF_C0 (NM_OFF_ZD + 64)
#define NM_OFF_G (NM_OFF_C0 + 4)                 // gate of the first layer [pair][unit], bf16 1.0 / 0.0
#define NM_OFF_GT (NM_OFF_G + 64 * NM_LG / 2)    // the same, [unit][pair]
#define NM_OFF_WH (NM_OFF_GT + 64 * NM_LG / 2)   // t2 / keep . W1 as [input][unit], bf16: the top 8 significant bits
#define NM_OFF_WM (NM_OFF_WH + 128 * NM_LG / 2)  // the next 8 (the last 8 stay in registers)
#define NM_LDS_FLOATS (NM_OFF_WM + 128 * NM_LG / 2)

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
  double inv_m;           // 1 / (B N), formed on the host: a kernel argument stays in scalar registers
  const int64_t* seed;    // device counter: the dropout stream of this call (NULL: 0)
  float *du_g, *du_m, *di_g, *di_m;
  float* raw;
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

// Row-chunk mapping of the gathers and the per-pair outputs: thread -> columns qc .. qc + 3 of KI pairs of a tile,
// qp, qp + 64 / KI, ... (a float4 of each of the pair's four rows); KI = 4 with four wavefronts, 2 with eight.
template <int KI>
struct NmIdx {
  int64_t u[KI], it[KI];
  unsigned have;          // bit k: the tile has a pair qp + (64 / KI) k
};

template <int KI>
__device__ __forceinline__ void nm_load_idx(const NmArgs& a, int64_t tile, int64_t tiles, int S, int N, int qp, NmIdx<KI>& x) {
  x.have = 0u;
#pragma unroll
  for (int k = 0; k < KI; ++k) { x.u[k] = 0; x.it[k] = 0; }
  if (tile >= tiles) return;
  const int64_t s0 = tile * S;
  const int ns = (int)((a.B - s0 < S) ? a.B - s0 : S);
  const int PT = ns * N;
#pragma unroll
  for (int k = 0; k < KI; ++k) {
    const int p = qp + (64 / KI) * k;
    if (p < PT) {
      x.u[k] = a.users[s0 + p / N];
      x.it[k] = a.items[s0 * N + p];
      x.have |= 1u << k;
    }
  }
}

// Row offsets of the fetch: a missing pair or an index outside its table reads row 0 (and is masked at use).
// Returns the bits of the pairs whose rows count.
template <int KI>
__device__ __forceinline__ unsigned nm_row_offsets(const NmArgs& a, const NmIdx<KI>& x, int64_t (&ou)[KI], int64_t (&oi)[KI], bool& bad) {
  unsigned ok = 0u;
#pragma unroll
  for (int k = 0; k < KI; ++k) {
    const int64_t u = x.u[k], it = x.it[k];
    const bool in = u >= 0 && u < a.n_users && it >= 0 && it < a.n_items;
    const bool have = (x.have >> k) & 1u;
    bad |= have && !in;
    ok |= (have && in) ? 1u << k : 0u;
    ou[k] = (in ? u : 0) * NM_D;
    oi[k] = (in ? it : 0) * NM_D;
  }
  return ok;
}

// Issues the four row loads of every pair of the mapping and returns at once.  The loads are unconditional so
// that nothing but the loads defines the registers: the wait for them then sits at their first use, a whole
// phase of matrix products later, where nm_mask_rows() zeroes what was not a pair.
template <int KI>
__device__ __forceinline__ void nm_load_rows(const NmArgs& a, const int64_t (&ou)[KI], const int64_t (&oi)[KI], int qc, float4 (&um)[KI], float4 (&im)[KI],
                                             float4 (&ug)[KI], float4 (&ig)[KI]) {
#pragma unroll
  for (int k = 0; k < KI; ++k) {
    um[k] = *reinterpret_cast<const float4*>(a.Um + ou[k] + qc);
    im[k] = *reinterpret_cast<const float4*>(a.Im + oi[k] + qc);
    ug[k] = *reinterpret_cast<const float4*>(a.Ug + ou[k] + qc);
    ig[k] = *reinterpret_cast<const float4*>(a.Ig + oi[k] + qc);
  }
}

template <int KI>
__device__ __forceinline__ void nm_mask_rows(unsigned ok, float4 (&um)[KI], float4 (&im)[KI], float4 (&ug)[KI], float4 (&ig)[KI]) {
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < KI; ++k)
    if (!((ok >> k) & 1u)) { um[k] = z; im[k] = z; ug[k] = z; ig[k] = z; }
}

// acc += A B over k = 0 .. 2 STEPS - 1 on v_mfma_f32_32x32x2_f32, operands in LDS: this lane's A element of step t
// at ap[2 t SA], its B element at bp[2 t SB] (both pointers already at the lane's index and k half).  The
// operands of the next eight steps are read while the eight products of the current ones run: with one
// wavefront per SIMD nothing else hides the LDS latency, and the scheduler left alone reads two operands,
// waits, and multiplies twice.
template <int STEPS, int SA, int SB, int BATCH = 8>
__device__ __forceinline__ void nm_mm(f32x16& acc, const float* ap, const float* bp) {
  float av[2][BATCH], bv[2][BATCH];
#pragma unroll
  for (int j = 0; j < BATCH; ++j) { av[0][j] = ap[2 * j * SA]; bv[0][j] = bp[2 * j * SB]; }
#pragma unroll
  for (int b = 0; b < STEPS / BATCH; ++b) {
    if (b + 1 < STEPS / BATCH) {
#pragma unroll
      for (int j = 0; j < BATCH; ++j) {
        av[(b + 1) & 1][j] = ap[2 * (BATCH * (b + 1) + j) * SA];
        bv[(b + 1) & 1][j] = bp[2 * (BATCH * (b + 1) + j) * SB];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < BATCH; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[b & 1][j], bv[b & 1][j], acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// two products with one A operand: acc1 += A B, acc2 += A B' with B' = B + BOFF floats
template <int STEPS, int SA, int SB, int BOFF>
__device__ __forceinline__ void nm_mm2(f32x16& acc1, f32x16& acc2, const float* ap, const float* bp) {
  float av[2][4], bv[2][4], cv[2][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { av[0][j] = ap[2 * j * SA]; bv[0][j] = bp[2 * j * SB]; cv[0][j] = bp[2 * j * SB + BOFF]; }
#pragma unroll
  for (int b = 0; b < STEPS / 4; ++b) {
    if (b + 1 < STEPS / 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        av[(b + 1) & 1][j] = ap[2 * (4 * (b + 1) + j) * SA];
        bv[(b + 1) & 1][j] = bp[2 * (4 * (b + 1) + j) * SB];
        cv[(b + 1) & 1][j] = bp[2 * (4 * (b + 1) + j) * SB + BOFF];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[b & 1][j], bv[b & 1][j], acc1, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[b & 1][j], cv[b & 1][j], acc2, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// WAVES = 4 or 8 wavefronts per workgroup (one workgroup per CU either way: the LDS map is the same).  With eight,
// every SIMD holds two wavefronts, so the scalar phases (staging, logits, outputs) run on twice the lanes and one
// wavefront's LDS waits hide behind the other's products; each wavefront then owns one 32 x 32 tile of dW1 and of dx
// instead of two, and the h1 product is split over k (the two halves meet in LDS).
// BCE-with-logits term of one pair and sigmoid(x), fp64.  Kept out of line: inlined into the tile loop, the two
// dozen fp64 coefficients of exp and log1p are held in registers across the whole loop (about fifty of them).
struct NmBce { double term, sig; };
__device__ __noinline__ NmBce nm_bce_term(double xd, double y) {
  const double e = exp(-fabs(xd));
  NmBce r;
  r.sig = xd >= 0.0 ? 1.0 / (1.0 + e) : e / (1.0 + e);
  r.term = fmax(xd, 0.0) - xd * y + log1p(e);
  return r;
}

// Eight fp32 values as three bf16 planes whose sum is the value, exactly: the top 8 significant bits, the next 8,
// the last 8 (each piece is cut, not rounded, so the remainders are exact and the third needs no cut).  A product
// with a 0 / 1 operand on v_mfma_f32_32x32x16_bf16 then accumulates exactly the fp32 terms, at an eighth of the
// passes of v_mfma_f32_32x32x2_f32.
__device__ __forceinline__ void nm_split8(const float (&v)[8], bf16x8& hi, bf16x8& mid, bf16x8& lo) {
  uint32_t h[8], m[8], l[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    h[k] = __float_as_uint(v[k]);
    const float r = v[k] - __uint_as_float(h[k] & 0xffff0000u);
    m[k] = __float_as_uint(r);
    l[k] = __float_as_uint(r - __uint_as_float(m[k] & 0xffff0000u));
  }
  uint32_t ph[4], pm[4], pl[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {     // element 2 j in the low half of the word: the upper halves of two floats, one byte permute
    ph[j] = __builtin_amdgcn_perm(h[2 * j + 1], h[2 * j], 0x07060302u);
    pm[j] = __builtin_amdgcn_perm(m[2 * j + 1], m[2 * j], 0x07060302u);
    pl[j] = __builtin_amdgcn_perm(l[2 * j + 1], l[2 * j], 0x07060302u);
  }
  hi = *reinterpret_cast<bf16x8*>(ph);
  mid = *reinterpret_cast<bf16x8*>(pm);
  lo = *reinterpret_cast<bf16x8*>(pl);
}

template <bool F64, int WAVES>
__global__ void __launch_bounds__(64 * WAVES)
neumf_step_kernel(NmArgs a) {
  constexpr int NT = 64 * WAVES;     // threads
  constexpr int KI = 1024 / NT;      // pairs per thread in the row-chunk mapping
  constexpr int QS = NT / 16;        // their stride
  constexpr int TP = NT / 64;        // threads per pair in the logit phase
  constexpr int PW = 64 / WAVES;     // pairs per wavefront in the dz1 phase
  extern __shared__ float lds[];
  float* X = lds + NM_OFF_X;       // tower input [pair][2 D]; its gradient once the products that read it are done
  float* H1 = lds + NM_OFF_H1;     // h1 [pair][D]; dz1 once the logit is known
  float* GU = lds + NM_OFF_GU;     // the pair's share of the user's GMF-row gradient
  float* W1T = lds + NM_OFF_W1T;   // W1 as [in][out]
  float* B1 = lds + NM_OFF_B1;
  float* T2 = lds + NM_OFF_T2;
  float* TG = lds + NM_OFF_TG;
  float* DL = lds + NM_OFF_DL;
  float* ZD = lds + NM_OFF_ZD;     // tg . zg of the pair
  float* C0 = lds + NM_OFF_C0;
  uint16_t* G16 = reinterpret_cast<uint16_t*>(lds + NM_OFF_G);
  uint16_t* GT16 = reinterpret_cast<uint16_t*>(lds + NM_OFF_GT);
  uint16_t* WH16 = reinterpret_cast<uint16_t*>(lds + NM_OFF_WH);
  uint16_t* WM16 = reinterpret_cast<uint16_t*>(lds + NM_OFF_WM);
  __shared__ double lsum_w[WAVES];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5, l31 = lane & 31;
  const int qc = (tid & 15) * 4, qp = tid >> 4;
  const int N = a.N;
  const int S = NM_P / N;                       // whole samples per tile
  const int64_t tiles = (a.B + S - 1) / S;

  // the first tile's rows are on their way while the weights are laid out
  NmIdx<KI> ix;
  float4 rum[KI], rim[KI], rug[KI], rig[KI];
  nm_load_idx(a, blockIdx.x, tiles, S, N, qp, ix);
  int64_t ou[KI], oi[KI];
  bool bad = false;
  unsigned rows_ok = nm_row_offsets(a, ix, ou, oi, bad);
  nm_load_rows(a, ou, oi, qc, rum, rim, rug, rig);

  for (int i = tid; i < 64 * 128; i += NT) W1T[(i & 127) * NM_LD + (i >> 7)] = a.W1[i];
  if (tid < 64) {
    float s = 0.f;
    for (int o = 0; o < 32; ++o) s += a.wf[32 + o] * a.W2[o * 64 + tid];
    T2[tid] = s;
    B1[tid] = a.b1[tid];
  } else if (tid < 128) {
    float s = 0.f;
    for (int o = 0; o < 32; ++o) s += a.wf[o] * a.Wg[o * 64 + (tid - 64)];
    TG[tid - 64] = s;
  } else if (tid == 128) {
    float s = a.bf[0];
    for (int o = 0; o < 32; ++o) s += a.wf[o] * a.bg[o];
    for (int o = 0; o < 32; ++o) s += a.wf[32 + o] * a.b2[o];
    C0[0] = s;
  }
  const uint64_t seed = a.seed ? (uint64_t)a.seed[0] : 0ull;
  const bool drop = a.p_drop > 0.f;
  const uint32_t thr = drop ? (uint32_t)((double)a.p_drop * 4294967296.0) : 0u;
  const float inv_keep = drop ? 1.f / (1.f - a.p_drop) : 1.f;

  // dW1 accumulators.  Four wavefronts: tiles (unit half wave % 2, input quarters 2 (wave / 2), + 1) of dz1^T x.
  // Eight: both unit halves of input quarter wave % 4 over the pairs 32 (wave / 4) .. + 31 of every tile, of
  // gate^T (dl x); the two pair halves meet, and t2[unit] / keep comes in, when the partial is written.
  f32x16 dW1a[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { dW1a[0][r] = 0.f; dW1a[1][r] = 0.f; }
  float va = 0.f, db1a = 0.f;       // column (tid % 64) of v and db1 over the pairs PW wave .. PW wave + PW - 1 of every tile
  float vga[4] = {0.f, 0.f, 0.f, 0.f};   // columns qc .. qc + 3 of vg over this thread's pairs
  float sdl = 0.f;                  // sum of dl over the pair tid / TP of every tile (threads with tid % TP == 0)
  double lsum = 0.0;
  __syncthreads();
  const float t2c = T2[tid & 63];
  const float4 tg4 = make_float4(TG[qc], TG[qc + 1], TG[qc + 2], TG[qc + 3]);
  const float c0 = C0[0];
  // eight wavefronts: the B operand of dx = dl (gate (t2 / keep . W1)) is the same for every tile -- three bf16
  // planes of 64 units x 32 inputs per wavefront (pair half wave % 2, input quarter wave / 2): the first two in
  // LDS (written by the wavefronts of pair half 0), the third in registers
  bf16x8 wlo[WAVES == 8 ? 4 : 1];
  if constexpr (WAVES == 8) {
    const int j = (wave >> 1) * 32 + l31;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      float w[8];
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const int u = 16 * ks + 8 * hh + kk;
        w[kk] = (T2[u] * inv_keep) * W1T[j * NM_LD + u];
      }
      bf16x8 whi, wmid;
      nm_split8(w, whi, wmid, wlo[ks]);
      if ((wave & 1) == 0) {
        *reinterpret_cast<bf16x8*>(WH16 + j * NM_LG + 16 * ks + 8 * hh) = whi;
        *reinterpret_cast<bf16x8*>(WM16 + j * NM_LG + 16 * ks + 8 * hh) = wmid;
      }
    }
  }

#ifdef NM_PHASE_TIMES
  long long stamp[9] = {};
#define NM_STAMP(i) if (tile == blockIdx.x + gridDim.x) stamp[i] = wall_clock64()
#else
#define NM_STAMP(i)
#endif
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    NM_STAMP(0);
    const int64_t s0 = tile * S;
    const int ns = (int)((a.B - s0 < S) ? a.B - s0 : S);
    const int PT = ns * N;
    const int64_t p0 = s0 * N;                  // first pair of the tile (pairs of a tile are contiguous)
    nm_load_idx(a, tile + gridDim.x, tiles, S, N, qp, ix);   // the next tile's indices: used after the logits
    // ---- 1. the fetched rows: tower input into LDS, tg . zg of every pair
    nm_mask_rows(rows_ok, rum, rim, rug, rig);
#pragma unroll
    for (int k = 0; k < KI; ++k) {
      const int p = qp + QS * k;
      float* xr = X + p * NM_LDX + qc;
      xr[0] = rum[k].x; xr[1] = rum[k].y; xr[2] = rum[k].z; xr[3] = rum[k].w;
      xr[64] = rim[k].x; xr[65] = rim[k].y; xr[66] = rim[k].z; xr[67] = rim[k].w;
      float zd = (rug[k].x * rig[k].x) * tg4.x + (rug[k].y * rig[k].y) * tg4.y + (rug[k].z * rig[k].z) * tg4.z + (rug[k].w * rig[k].w) * tg4.w;
      zd = rx_group_sum(zd, 16);
      if ((tid & 15) == 0) ZD[p] = zd;
    }
    __syncthreads();
    NM_STAMP(1);
    // ---- 2. h1 = dropout(relu(x W1^T + b1)): wave -> 32 x 32 tile (rt, ct) of [64 pairs, 64 units]; with eight
    //         wavefronts, half kh of the 128 inputs each: the upper halves leave their sums in h1's place
    {
      const int w4 = wave & 3, kh = wave >> 2;
      const int rt = w4 >> 1, ct = w4 & 1;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const float* ap = X + (rt * 32 + l31) * NM_LDX + hh;
      const float* bp = W1T + hh * NM_LD + ct * 32 + l31;
      const int col = ct * 32 + l31;
      if (WAVES == 4) {
        nm_mm<64, 1, NM_LD>(acc, ap, bp);
      } else {
        nm_mm<32, 1, NM_LD, 4>(acc, ap + kh * 64, bp + kh * 64 * NM_LD);
        if (kh == 1) {
#pragma unroll
          for (int r = 0; r < 16; ++r) H1[(rt * 32 + NM_ROW(r, hh)) * NM_LD + col] = acc[r];
        }
        __syncthreads();
      }
      const float bv = B1[col];
      if (WAVES == 4 || kh == 0) {
      uint16_t gate[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rt * 32 + NM_ROW(r, hh);
        if (WAVES == 8) acc[r] += H1[row * NM_LD + col];
        float v = acc[r] + bv;
        v = v > 0.f ? v : 0.f;
        if (drop) {
          int rowv = row;
          asm volatile("" : "+v"(rowv));    // (formed here: sixteen 64-bit counters kept across the tile loop would cost 32 registers)
          const uint32_t draw = (uint32_t)(nm_mix(seed, (uint64_t)(p0 + rowv) * 64ull + (uint64_t)col) >> 32);
          v = draw >= thr ? v * inv_keep : 0.f;
        }
        v = row < PT ? v : 0.f;
        H1[row * NM_LD + col] = v;
        if constexpr (WAVES == 8) {
          gate[r] = v > 0.f ? (uint16_t)0x3F80 : (uint16_t)0;
          G16[row * NM_LG + col] = gate[r];
        }
      }
      if constexpr (WAVES == 8) {
#pragma unroll
        for (int q = 0; q < 4; ++q)     // rows 8 q + 4 hh .. + 3 of the tile sit in registers 4 q .. 4 q + 3
          *reinterpret_cast<uint2*>(GT16 + col * NM_LG + rt * 32 + 8 * q + 4 * hh) =
              make_uint2((uint32_t)gate[4 * q] | ((uint32_t)gate[4 * q + 1] << 16), (uint32_t)gate[4 * q + 2] | ((uint32_t)gate[4 * q + 3] << 16));
      }
      }
    }
    __syncthreads();
    NM_STAMP(2);
    // ---- 3. logit = c0 + tg . zg + t2 . h1 (TP threads per pair), BCE term (fp64), dL/dlogit
    // The next tile's row offsets are taken here, pinned in registers: the indices arrived two phases ago, and
    // a wait placed behind the label loads and gradient stores below would wait for those as well.
    rows_ok = nm_row_offsets(a, ix, ou, oi, bad);
#pragma unroll
    for (int k = 0; k < KI; ++k) asm volatile("" : "+v"(ou[k]), "+v"(oi[k]));
    {
      int p = tid / TP;
      asm volatile("" : "+v"(p));     // (addresses off p are formed here, per tile: kept across the loop they cost registers all the way)
      const int part = (tid % TP) * (64 / TP);
      const float* hp = H1 + p * NM_LD + part;
      float x = 0.f;
#pragma unroll
      for (int j = 0; j < 64 / TP; ++j) x += hp[j] * T2[part + j];
      x = rx_group_sum(x, TP);
      if (tid % TP == 0) {
        float dl = 0.f;
        if (p < PT) {
          x += c0 + ZD[p];
          if (a.logits) a.logits[p0 + p] = x;
          const double xd = (double)x;
          const double y = F64 ? reinterpret_cast<const double*>(a.labels)[p0 + p] : (double)reinterpret_cast<const float*>(a.labels)[p0 + p];
          const NmBce t = nm_bce_term(xd, y);
          lsum += t.term;
          dl = (float)((t.sig - y) * a.inv_m);
        }
        DL[p] = dl;
        sdl += dl;
      }
    }
    __syncthreads();
    NM_STAMP(3);
    // ---- 4. v += dl h1, dz1 = dl t2 through the dropout scale and the ReLU (in place), db1 += dz1;
    //         vg += dl zg, item GMF gradients out, the pair's share of the user's GMF gradient into LDS
    {
      const int c = tid & 63;
#pragma unroll
      for (int p = wave * PW; p < wave * PW + PW; ++p) {
        const float h = H1[p * NM_LD + c], dl = DL[p];
        va += dl * h;
        if constexpr (WAVES == 8) {
          db1a += h > 0.f ? dl : 0.f;      // (t2 / keep comes in at the end)
        } else {
          const float dz = h > 0.f ? (dl * t2c) * inv_keep : 0.f;
          db1a += dz;
          H1[p * NM_LD + c] = dz;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < KI; ++k) {
      const int p = qp + QS * k;
      const float dl = DL[p];
      vga[0] += dl * (rug[k].x * rig[k].x); vga[1] += dl * (rug[k].y * rig[k].y);
      vga[2] += dl * (rug[k].z * rig[k].z); vga[3] += dl * (rug[k].w * rig[k].w);
      const float4 dz = make_float4(dl * tg4.x, dl * tg4.y, dl * tg4.z, dl * tg4.w);
      if (p < PT) *reinterpret_cast<float4*>(a.di_g + (p0 + p) * NM_D + qc) = make_float4(dz.x * rug[k].x, dz.y * rug[k].y, dz.z * rug[k].z, dz.w * rug[k].w);
      float* gr = GU + p * NM_LD + qc;
      gr[0] = dz.x * rig[k].x; gr[1] = dz.y * rig[k].y; gr[2] = dz.z * rig[k].z; gr[3] = dz.w * rig[k].w;
    }
    // the next tile's rows travel while the gradient products below run
    nm_load_rows(a, ou, oi, qc, rum, rim, rug, rig);
    __syncthreads();
    NM_STAMP(4);
    // ---- 5. dW1 += dz1^T x; dx = dz1 W1 (two 32 x 32 tiles of each per wave with four wavefronts, one with eight)
    if constexpr (WAVES == 8) {
      // The gate is 0 / 1, so dz1 = gate (dl t2 / keep) factors: dW1 = (t2 / keep) . gate^T (dl x) and
      // dx = dl . gate (t2 / keep . W1), with the fp32 factor cut into three bf16 planes (nm_split8).
      {
        const int cq = wave & 3, kh = wave >> 2;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const int k0 = kh * 32 + ks * 16 + 8 * hh;       // this lane's eight pairs
          float xv[8];
#pragma unroll
          for (int kk = 0; kk < 8; ++kk) xv[kk] = X[(k0 + kk) * NM_LDX + cq * 32 + l31] * DL[k0 + kk];
          bf16x8 bh, bm, bl;
          nm_split8(xv, bh, bm, bl);
#pragma unroll
          for (int ht = 0; ht < 2; ++ht) {
            const bf16x8 ga = *reinterpret_cast<const bf16x8*>(GT16 + (ht * 32 + l31) * NM_LG + k0);
            dW1a[ht] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga, bh, dW1a[ht], 0, 0, 0);
            dW1a[ht] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga, bm, dW1a[ht], 0, 0, 0);
            dW1a[ht] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga, bl, dW1a[ht], 0, 0, 0);
          }
        }
      }
      f32x16 acc1;
      const int rt = wave & 1, cq = wave >> 1;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 ga = *reinterpret_cast<const bf16x8*>(G16 + (rt * 32 + l31) * NM_LG + 16 * ks + 8 * hh);
        const bf16x8 wh = *reinterpret_cast<const bf16x8*>(WH16 + (cq * 32 + l31) * NM_LG + 16 * ks + 8 * hh);
        const bf16x8 wm = *reinterpret_cast<const bf16x8*>(WM16 + (cq * 32 + l31) * NM_LG + 16 * ks + 8 * hh);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga, wh, acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga, wm, acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga, wlo[ks], acc1, 0, 0, 0);
      }
      NM_STAMP(5);
      __syncthreads();    // every read of x is done: it now takes dx
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rt * 32 + NM_ROW(r, hh);
        X[row * NM_LDX + cq * 32 + l31] = acc1[r] * DL[row];
      }
    } else {
      f32x16 acc1, acc2;
      const int ht = wave & 1, cp = wave >> 1;
      const float* ap = H1 + hh * NM_LD + ht * 32 + l31;                 // A[i = unit][k = pair]
      const float* bp = X + hh * NM_LDX + cp * 64 + l31;                 // B[k = pair][j = input]
      nm_mm2<NM_P / 2, NM_LD, NM_LDX, 32>(dW1a[0], dW1a[1], ap, bp);
      const int rt = wave & 1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc1[r] = 0.f; acc2[r] = 0.f; }
      const float* a1 = H1 + (rt * 32 + l31) * NM_LD + hh;               // A[i = pair][k = unit]
      const float* b1p = W1T + (cp * 64 + l31) * NM_LD + hh;             // B[k = unit][j = input] = W1[unit][input]
      nm_mm2<32, 1, 1, 32 * NM_LD>(acc1, acc2, a1, b1p);
      NM_STAMP(5);
      __syncthreads();    // every read of x is done: it now takes dx
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rt * 32 + NM_ROW(r, hh);
        X[row * NM_LDX + cp * 64 + l31] = acc1[r];
        X[row * NM_LDX + cp * 64 + 32 + l31] = acc2[r];
      }
    }
    __syncthreads();
    NM_STAMP(6);
    // ---- 6. embedding gradients out: item MLP rows per pair, user rows summed over the sample's pairs in pair order
#pragma unroll
    for (int k = 0; k < KI; ++k) {
      const int p = qp + QS * k;
      if (p < PT) {
        const float* xr = X + p * NM_LDX + 64 + qc;
        *reinterpret_cast<float4*>(a.di_m + (p0 + p) * NM_D + qc) = make_float4(xr[0], xr[1], xr[2], xr[3]);
      }
    }
    for (int q = tid; q < ns * 16; q += NT) {
      const int s = q >> 4, c = (q & 15) * 4;
      float4 m = make_float4(0.f, 0.f, 0.f, 0.f), g = m;
      for (int n = 0; n < N; ++n) {
        const float* xr = X + (s * N + n) * NM_LDX + c;
        const float* gr = GU + (s * N + n) * NM_LD + c;
        m.x += xr[0]; m.y += xr[1]; m.z += xr[2]; m.w += xr[3];
        g.x += gr[0]; g.y += gr[1]; g.z += gr[2]; g.w += gr[3];
      }
      *reinterpret_cast<float4*>(a.du_m + (s0 + s) * NM_D + c) = m;
      *reinterpret_cast<float4*>(a.du_g + (s0 + s) * NM_D + c) = g;
    }
    __syncthreads();
    NM_STAMP(7);
  }
#ifdef NM_PHASE_TIMES
  if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == 100))
    printf("block %d phases (10 ns ticks): stage %lld  h1 %lld  logit %lld  dz %lld  mfma %lld  dx %lld  out %lld  | tile %lld\n", (int)blockIdx.x,
           stamp[1] - stamp[0], stamp[2] - stamp[1], stamp[3] - stamp[2], stamp[4] - stamp[3], stamp[5] - stamp[4], stamp[6] - stamp[5],
           stamp[7] - stamp[6], stamp[7] - stamp[0]);
#endif

  if (bad && a.err_flag) atomicOr(a.err_flag, RX_ERR_INDEX);
  // ---- this workgroup's partial: [dW1 | db1 | v | vg | sum dl]; the thread-level pieces meet in LDS, fixed order
  float* wp = a.raw + (size_t)blockIdx.x * NM_RAW;
  if constexpr (WAVES == 8) {
    // the two pair halves of gate^T (dl x) meet in LDS (x is done with); the unit's t2 / keep comes in here
    const int cq = wave & 3, kh = wave >> 2;
    float* half = X + cq * 2048;
    if (kh == 1) {
#pragma unroll
      for (int ht = 0; ht < 2; ++ht)
#pragma unroll
        for (int r = 0; r < 16; ++r) half[(ht * 16 + r) * 64 + lane] = dW1a[ht][r];
    }
    __syncthreads();
    if (kh == 0) {
#pragma unroll
      for (int ht = 0; ht < 2; ++ht)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int u = ht * 32 + NM_ROW(r, hh);
          wp[u * 128 + cq * 32 + l31] = (dW1a[ht][r] + half[(ht * 16 + r) * 64 + lane]) * (T2[u] * inv_keep);
        }
    }
    __syncthreads();
    db1a *= t2c * inv_keep;
  } else {
    const int ht = wave & 1, cp = wave >> 1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int h = ht * 32 + NM_ROW(r, hh);
      wp[h * 128 + cp * 64 + l31] = dW1a[0][r];
      wp[h * 128 + cp * 64 + 32 + l31] = dW1a[1][r];
    }
  }
  float* red = X;        // [db1: WAVES x 64 | v: WAVES x 64 at 512 | vg: QS x 64 at 1024 | sum dl: 64 at 3072]
  red[wave * 64 + lane] = db1a;
  red[512 + wave * 64 + lane] = va;
#pragma unroll
  for (int e = 0; e < 4; ++e) red[1024 + qp * 64 + qc + e] = vga[e];
  if (tid % TP == 0) red[3072 + tid / TP] = sdl;
  lsum = rx_wave_sum_f64(lsum);
  if (lane == 0) lsum_w[wave] = lsum;
  __syncthreads();
  if (tid < 64) {
    float s = 0.f, t = 0.f;
#pragma unroll
    for (int g = 0; g < WAVES; ++g) { s += red[g * 64 + tid]; t += red[512 + g * 64 + tid]; }
    wp[64 * 128 + tid] = s;
    wp[NM_RAW_V + tid] = t;
  } else if (tid < 128) {
    float s = 0.f;
    for (int g = 0; g < QS; ++g) s += red[1024 + g * 64 + (tid - 64)];
    wp[NM_RAW_VG + (tid - 64)] = s;
  } else if (tid < 192) {
    const float s = rx_wave_sum(red[3072 + (tid - 128)]);
    if (tid == 128) wp[NM_RAW_SDL] = s;
  }
  if (tid == 0) {
    double l = 0.0;
#pragma unroll
    for (int g = 0; g < WAVES; ++g) l += lsum_w[g];
    a.loss_part[blockIdx.x] = l;
  }
}

// Adds the workgroups' partials in a fixed order (bitwise reproducible), expands v, vg and sum(dl) into the
// gradients of the layers behind the first tower layer, finishes the loss and advances the dropout counter.
struct NmFin {
  const float* raw;
  int n_parts;
  float* wgrad;           // [W1 | b1 | W2 | b2 | Wg | bg | wf | bf]
  const float *W2, *b2, *Wg, *bg, *wf;
  const double* loss_part;
  double count;
  double* loss;
  int64_t* seed;
};

#define NM_FB 1024
#define NM_FIN_W1_BLOCKS ((64 * 128 + 64) / 64)     // 129 blocks of 64 columns: dW1 and db1 keep their places
#define NM_FIN_BLOCKS (NM_FIN_W1_BLOCKS + 2 + 2 + 1)
#define NM_WG_W2 (64 * 128 + 64)
#define NM_WG_B2 (NM_WG_W2 + 32 * 64)
#define NM_WG_WG (NM_WG_B2 + 32)
#define NM_WG_BG (NM_WG_WG + 32 * 64)
#define NM_WG_WF (NM_WG_BG + 32)
#define NM_WG_BF (NM_WG_WF + 64)

// columns col0 .. col0 + 63 of the partials: 16 thread groups take n_parts / 16 partials each, then meet in order
__device__ __forceinline__ void nm_reduce64(const float* __restrict__ raw, int n_parts, int col0, float* part, float* out) {
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int chunk = (n_parts + 15) >> 4;        // at most NM_MAX_GRID / 16 = 16 partials: all loads in flight at once
  const int b0 = g * chunk;
  float v[NM_MAX_GRID / 16];
#pragma unroll
  for (int k = 0; k < NM_MAX_GRID / 16; ++k) {
    const int b = b0 + k;
    v[k] = (k < chunk && b < n_parts) ? raw[(size_t)b * NM_RAW + col0 + c] : 0.f;
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NM_MAX_GRID / 16; ++k) s += v[k];
  part[g * 64 + c] = s;
  __syncthreads();
  if (threadIdx.x < 64) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += part[k * 64 + c];
    out[c] = t;
  }
  __syncthreads();
}

__global__ void __launch_bounds__(NM_FB)
neumf_finish_kernel(NmFin f) {
  __shared__ float part[16 * 64];
  __shared__ float outv[64], outg[64];
  const int tid = threadIdx.x, b = blockIdx.x;
  if (b < NM_FIN_W1_BLOCKS) {
    nm_reduce64(f.raw, f.n_parts, b * 64, part, outv);
    if (tid < 64) f.wgrad[b * 64 + tid] = outv[tid];
  } else if (b < NM_FIN_W1_BLOCKS + 2) {           // dW2[o][c] = wf[D/2 + o] v[c]
    nm_reduce64(f.raw, f.n_parts, NM_RAW_V, part, outv);
    const int idx = (b - NM_FIN_W1_BLOCKS) * NM_FB + tid;
    f.wgrad[NM_WG_W2 + idx] = f.wf[32 + (idx >> 6)] * outv[idx & 63];
  } else if (b < NM_FIN_W1_BLOCKS + 4) {           // dWg[o][c] = wf[o] vg[c]
    nm_reduce64(f.raw, f.n_parts, NM_RAW_VG, part, outv);
    const int idx = (b - NM_FIN_W1_BLOCKS - 2) * NM_FB + tid;
    f.wgrad[NM_WG_WG + idx] = f.wf[idx >> 6] * outv[idx & 63];
  } else {
    // the last block: sum(dl) and the loss over the partials (one load per thread, fixed trees), v and vg, then
    // db2, dbg, dbf and dwf = [Wg vg + bg sum(dl) | W2 v + b2 sum(dl)] (16 lanes per output, 16-byte loads)
    __shared__ float sdl_w[4];
    __shared__ double loss_w[4];
    if (tid < 256) {
      const float x = tid < f.n_parts ? f.raw[(size_t)tid * NM_RAW + NM_RAW_SDL] : 0.f;
      const double l = tid < f.n_parts ? f.loss_part[tid] : 0.0;
      const float xs = rx_wave_sum(x);
      const double ls = rx_wave_sum_f64(l);
      if ((tid & 63) == 0) { sdl_w[tid >> 6] = xs; loss_w[tid >> 6] = ls; }
    }
    nm_reduce64(f.raw, f.n_parts, NM_RAW_V, part, outv);      // (its barriers also publish sdl_w / loss_w)
    nm_reduce64(f.raw, f.n_parts, NM_RAW_VG, part, outg);
    const float sdl = (sdl_w[0] + sdl_w[1]) + (sdl_w[2] + sdl_w[3]);
    if (tid == 0) {
      f.loss[0] = ((loss_w[0] + loss_w[1]) + (loss_w[2] + loss_w[3])) / f.count;
      if (f.seed) f.seed[0] = f.seed[0] + 1;
      f.wgrad[NM_WG_BF] = sdl;
    }
    if (tid < 32) {
      f.wgrad[NM_WG_B2 + tid] = f.wf[32 + tid] * sdl;
      f.wgrad[NM_WG_BG + tid] = f.wf[tid] * sdl;
    }
    {
      const int j = tid >> 4, q = (tid & 15) * 4, o = j & 31;
      const float4 w = *reinterpret_cast<const float4*>((j < 32 ? f.Wg : f.W2) + o * 64 + q);
      const float* x = (j < 32 ? outg : outv) + q;
      float s = ((w.x * x[0] + w.y * x[1]) + w.z * x[2]) + w.w * x[3];
      s = rx_group_sum(s, 16);
      if ((tid & 15) == 0) f.wgrad[NM_WG_WF + j] = s + (j < 32 ? f.bg[o] : f.b2[o]) * sdl;
    }
  }
}

static int nm_grid(int64_t B, int N) {
  const int S = NM_P / N;
  const int64_t tiles = (B + S - 1) / S;
  // every workgroup takes the same number of tiles: the launch is as long as its busiest compute unit either
  // way, and the units it leaves free run whatever else is in flight (the index preparation of the next step)
  const int64_t rounds = (tiles + NM_MAX_GRID - 1) / NM_MAX_GRID;
  return (int)((tiles + rounds - 1) / rounds);
}

extern "C" int rx_neumf_step_supported(int32_t dim, int32_t N) { return dim == NM_D && N >= 1 && N <= NM_P; }

extern "C" int64_t rx_neumf_step_weight_floats(void) { return NM_WP; }

extern "C" size_t rx_neumf_step_workspace_bytes(int64_t B, int32_t N) {
  if (B < 1 || N < 1 || N > NM_P) return 256;
  const int g = nm_grid(B, N);
  return rx_align_up((size_t)g * NM_RAW * sizeof(float), 256) + rx_align_up((size_t)g * sizeof(double), 256);
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
  RX_REQUIRE(rx_aligned16(r->w2) && rx_aligned16(r->wg), "rx_neumf_step: linear_mlp.3 / linear_gmf weights must be 16-byte aligned");
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
  a.inv_m = 1.0 / ((double)r->B * (double)r->N);
  a.du_g = r->du_gmf; a.du_m = r->du_mlp; a.di_g = r->di_gmf; a.di_m = r->di_mlp;
  a.raw = (float*)ws;
  a.loss_part = (double*)((char*)ws + rx_align_up((size_t)grid * NM_RAW * sizeof(float), 256));
  a.logits = r->logits; a.err_flag = err_flag;
  const size_t lds = NM_LDS_FLOATS * sizeof(float);
  // RX_NEUMF_WAVES=4 runs the step with four wavefronts per workgroup instead of eight (A/B timing)
  static const bool four = [] { const char* e = getenv("RX_NEUMF_WAVES"); return e && e[0] == '4'; }();
  static bool attr_set[64] = {};
  if (device >= 64 || !attr_set[device]) {   // LDS beyond 64 KiB per workgroup has to be asked for, per device
    RX_HIP(hipFuncSetAttribute((const void*)neumf_step_kernel<true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    RX_HIP(hipFuncSetAttribute((const void*)neumf_step_kernel<false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    RX_HIP(hipFuncSetAttribute((const void*)neumf_step_kernel<true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    RX_HIP(hipFuncSetAttribute((const void*)neumf_step_kernel<false, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (device < 64) attr_set[device] = true;
  }
  hipStream_t st = (hipStream_t)stream;
  if (four) {
    if (r->labels_f64) hipLaunchKernelGGL((neumf_step_kernel<true, 4>), dim3(grid), dim3(256), lds, st, a);
    else hipLaunchKernelGGL((neumf_step_kernel<false, 4>), dim3(grid), dim3(256), lds, st, a);
  } else {
    if (r->labels_f64) hipLaunchKernelGGL((neumf_step_kernel<true, 8>), dim3(grid), dim3(512), lds, st, a);
    else hipLaunchKernelGGL((neumf_step_kernel<false, 8>), dim3(grid), dim3(512), lds, st, a);
  }
  RX_LAUNCH_CHECK();
  NmFin f{};
  f.raw = a.raw; f.n_parts = grid; f.wgrad = r->wgrad;
  f.W2 = r->w2; f.b2 = r->b2; f.Wg = r->wg; f.bg = r->bg; f.wf = r->wf;
  f.loss_part = a.loss_part; f.count = (double)r->B * (double)r->N; f.loss = r->loss; f.seed = const_cast<int64_t*>(r->seed);
  hipLaunchKernelGGL(neumf_finish_kernel, dim3(NM_FIN_BLOCKS), dim3(NM_FB), 0, st, f);
  RX_LAUNCH_CHECK();
  return 0;
}
