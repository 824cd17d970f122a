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
// which leaves three matrix products per pair tile: the first tower layer forward, and its two gradients
// dW1 += dz1^T x and dx = dz1 W1.  fp32 matrix instructions (v_mfma_f32_32x32x2_f32) run at a sixteenth of the bf16
// rate, and these three products were two thirds of the tile's time on them, so they run on
// v_mfma_f32_32x32x16_bf16 with every fp32 operand cut into three bf16 planes that add up to it exactly (nm_split8):
//   * dz1 = gate . (dl t2 / keep) with gate = [h1 > 0] in {0, 1}: dW1 = (t2 / keep) . gate^T (dl x) and
//     dx = dl . gate (t2 / keep . W1).  One operand is 0 / 1, the other three planes: every product is exact, the
//     sum is the fp32 accumulation of the fp32 terms.
//   * h1 = x W1^T: of the nine plane products the six of weight >= 2^-16 are formed; what is left out is below the
//     rounding of a single fp32 product.  (A unit whose pre-activation is zero to rounding may fall on the other
//     side of the ReLU than under another summation order, as with any reordering of the sum.)
// The only sums across pairs are the weight-gradient terms, the gradient of a user's rows over that user's N
// items, and the loss.  A workgroup takes tiles of whole samples (64 / N samples, at most 64 pairs), keeps the
// tile's activations in LDS, keeps gate^T (dl x) in accumulator registers across its tiles, fetches the rows of
// its next tile while the gradient products of the current one run, and leaves
//   du_g, du_m [B, D]      gradient of the user's GMF / MLP row, summed over the sample's N pairs (pair order)
//   di_g, di_m [B N, D]    gradient of every pair's item rows
//   raw [grid][NM_RAW]     per-workgroup [dW1 | db1 | v | vg | sum(dl)], added in workgroup order by the finish
//                          kernel, which also expands v, vg, sum(dl) into the gradients of W2, b2, Wg, bg, wf, bf
//   loss partials          fp64, one per workgroup
// The embedding gradients then take the usual deterministic path (sort + segmented reduce, fused row update).
// Written for D = 64 (tables of 64 floats, tower 128 -> 64 -> 32, GMF 64 -> 32): 155 KB of LDS per workgroup, eight
// wavefronts, one workgroup per CU.
//
// LDS: the fp32 arrays (x, h1) have an ODD row pitch, so a wavefront's reads of 32 rows at one column, and of 32
// columns of one row, both spread over all banks; the bf16 arrays (W1 planes, the gate in both orientations, the
// planes of t2 / keep . W1) are read 16 bytes per lane and have row pitches of 36 and 68 words for the same reason.
#include "rx_common.h"

#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

#define NM_D 64
#define NM_P 64       // pair rows of a tile (two MFMA row tiles)
#define NM_LDX 129    // pitch of the tower input / its gradient [pair][2 D]
#define NM_LD 65      // pitch of the 64-wide arrays
#define NM_LG 72      // pitch (bf16 elements) of the 64-wide bf16 arrays: 16-byte reads of 32 consecutive rows touch every bank once
#define NM_LW 136     // pitch (bf16 elements) of the planes of W1 [unit][input]: likewise
#define NM_WP (64 * 128 + 64 + 32 * 64 + 32 + 32 * 64 + 32 + 64 + 1)   // floats of the weight gradient [W1|b1|W2|b2|Wg|bg|wf|bf]
#define NM_RAW (64 * 128 + 64 + 64 + 64 + 1)                            // floats of one workgroup's partial [dW1|db1|v|vg|sum dl]
#define NM_RAW_V (64 * 128 + 64)
#define NM_RAW_VG (NM_RAW_V + 64)
#define NM_RAW_SDL (NM_RAW_VG + 64)
#define NM_MAX_GRID 256

// LDS map (floats)
#define NM_THREADS 512
#define NM_OFF_X 0
#define NM_OFF_H1 (NM_OFF_X + NM_P * NM_LDX)
#define NM_OFF_W1P (NM_OFF_H1 + NM_P * NM_LD)              // W1 [unit][input] as three bf16 planes
#define NM_OFF_B1 (NM_OFF_W1P + 3 * 64 * NM_LW / 2)
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
static_assert(NM_OFF_H1 % 4 == 0 && NM_OFF_W1P % 4 == 0 && NM_OFF_G % 4 == 0 && NM_OFF_GT % 4 == 0 && NM_OFF_WH % 4 == 0 && NM_OFF_WM % 4 == 0,
              "16-byte reads of the bf16 arrays");
static_assert(NM_LDS_FLOATS * 4 <= 160 * 1024, "LDS of one CU");

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
// the last 8 (each piece is cut, not rounded, so the remainders are exact and the third needs no cut).
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

// Sum over aligned groups of 8 / 16 lanes, every lane of the group getting it, on the data-parallel-primitive
// path of the vector unit (lane exchanges inside a row of 16: no LDS crossbar round trip per step).  After the two
// quad steps the four lanes of a quad hold one value, so the mirrors stand in for the exchanges at distance 4 / 8.
template <int CTRL>
__device__ __forceinline__ float nm_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float nm_sum8(float v) {
  v += nm_dpp<0xB1>(v);      // quad_perm [1, 0, 3, 2]
  v += nm_dpp<0x4E>(v);      // quad_perm [2, 3, 0, 1]
  v += nm_dpp<0x141>(v);     // row_half_mirror
  return v;
}
__device__ __forceinline__ float nm_sum16(float v) {
  v = nm_sum8(v);
  v += nm_dpp<0x140>(v);     // row_mirror
  return v;
}

#define NM_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

// Eight wavefronts per workgroup, one workgroup per CU (LDS): every SIMD holds two wavefronts, so one's operand
// preparation (LDS reads, plane cuts) runs under the other's matrix instructions.
template <bool F64>
__global__ void __launch_bounds__(NM_THREADS)
neumf_step_kernel(NmArgs a) {
  constexpr int NT = NM_THREADS;
  constexpr int WAVES = NT / 64;
  constexpr int KI = 1024 / NT;      // pairs per thread in the row-chunk mapping
  constexpr int QS = NT / 16;        // their stride
  constexpr int TP = NT / 64;        // threads per pair in the logit phase
  constexpr int PW = 64 / WAVES;     // pairs per wavefront in the v / db1 phase
  constexpr int PL = 64 * NM_LW;     // bf16 elements of one plane of W1
  extern __shared__ __align__(16) float lds[];
  float* X = lds + NM_OFF_X;       // tower input [pair][2 D]; its gradient once the products that read it are done
  float* H1 = lds + NM_OFF_H1;     // h1 [pair][D]
  float* GU = lds + NM_OFF_H1;     // ... and, once h1 is done with, the pair's share of the user's GMF-row gradient
  uint16_t* WP16 = reinterpret_cast<uint16_t*>(lds + NM_OFF_W1P);   // W1 [unit][input], three bf16 planes
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

  for (int c = tid; c < 64 * 16; c += NT) {       // W1 [unit][input] in pieces of eight inputs
    const int u = c >> 4, k0 = (c & 15) * 8;
    float w[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) w[kk] = a.W1[u * 128 + k0 + kk];
    bf16x8 wh, wm, wl;
    nm_split8(w, wh, wm, wl);
    *reinterpret_cast<bf16x8*>(WP16 + u * NM_LW + k0) = wh;
    *reinterpret_cast<bf16x8*>(WP16 + PL + u * NM_LW + k0) = wm;
    *reinterpret_cast<bf16x8*>(WP16 + 2 * PL + u * NM_LW + k0) = wl;
  }
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

  // gate^T (dl x) accumulated over this workgroup's tiles: both unit halves of input quarter wave % 4 over the
  // pairs 32 (wave / 4) .. + 31 of every tile; the two pair halves meet, and t2[unit] / keep comes in, when the
  // partial is written
  f32x16 dW1a[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { dW1a[0][r] = 0.f; dW1a[1][r] = 0.f; }
  float va = 0.f, db1a = 0.f;       // column (tid % 64) of v and db1 over the pairs PW wave .. PW wave + PW - 1 of every tile
  float vga[4] = {0.f, 0.f, 0.f, 0.f};   // columns qc .. qc + 3 of vg over this thread's pairs
  float sdl = 0.f;                  // first wavefront: sum of dl over the pair `lane` of every tile
  double lsum = 0.0;
  __syncthreads();
  const float t2c = T2[tid & 63];
  const float4 tg4 = make_float4(TG[qc], TG[qc + 1], TG[qc + 2], TG[qc + 3]);
  const float c0 = C0[0];
  // The B operand of dx = dl (gate (t2 / keep . W1)) is the same for every tile: three bf16 planes of 64 units x 32
  // inputs per wavefront (pair half wave % 2, input quarter wave / 2), the first two in LDS (written by the
  // wavefronts of pair half 0), the third in registers.
  bf16x8 wlo[4];
  {
    const int j = (wave >> 1) * 32 + l31;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      float w[8];
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const int u = 16 * ks + 8 * hh + kk;
        w[kk] = (T2[u] * inv_keep) * a.W1[u * 128 + j];
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
    // ---- 1. the fetched rows: tower input into LDS, tg . zg of every pair
    nm_mask_rows(rows_ok, rum, rim, rug, rig);
#ifdef NM_PHASE_TIMES
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    NM_STAMP(8);
#endif
#pragma unroll
    for (int k = 0; k < KI; ++k) {
      const int p = qp + QS * k;
      float* xr = X + p * NM_LDX + qc;
      xr[0] = rum[k].x; xr[1] = rum[k].y; xr[2] = rum[k].z; xr[3] = rum[k].w;
      xr[64] = rim[k].x; xr[65] = rim[k].y; xr[66] = rim[k].z; xr[67] = rim[k].w;
      float zd = (rug[k].x * rig[k].x) * tg4.x + (rug[k].y * rig[k].y) * tg4.y + (rug[k].z * rig[k].z) * tg4.z + (rug[k].w * rig[k].w) * tg4.w;
      zd = nm_sum16(zd);
      if ((tid & 15) == 0) ZD[p] = zd;
    }
    // (issued only now: a wait for the fetched rows above also waits for every load issued before it)
    {
      int qpv = qp;
      asm volatile("" : "+v"(qpv));      // (index addresses are formed here, per tile: 64-bit pieces of them kept across the loop cost registers)
      nm_load_idx(a, tile + gridDim.x, tiles, S, N, qpv, ix);   // the next tile's indices: used after the logits
    }
    // the label of pair `lane` (used by the first wavefront), on its way until the logit is known: loaded by every
    // wavefront from a clamped address and left as it is -- a branch around the load, or a conversion behind it,
    // would make the tile wait for it here
    typename std::conditional<F64, double, float>::type yraw;
    {
      int pl = lane < PT ? lane : PT - 1;
      asm volatile("" : "+v"(pl));     // (the address is formed here, per tile)
      yraw = reinterpret_cast<const typename std::conditional<F64, double, float>::type*>(a.labels)[p0 + pl];
    }
    __syncthreads();
    NM_STAMP(1);
    // ---- 2. h1 = dropout(relu(x W1^T + b1)): wave -> 32 x 32 tile (rt, ct) of [64 pairs, 64 units] over half kh of
    //         the 128 inputs; the upper halves leave their sums in h1's place, the lower halves finish.  x is cut
    //         into planes as it is read; of the nine plane products the six of weight >= 2^-16 are formed, smallest
    //         first (what is left out is below the rounding of one fp32 product).
    {
      const int w4 = wave & 3, kh = wave >> 2;
      const int rt = w4 >> 1, ct = w4 & 1;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const float* xrow = X + (rt * 32 + l31) * NM_LDX + kh * 64 + 8 * hh;
      const uint16_t* wrow = WP16 + (ct * 32 + l31) * NM_LW + kh * 64 + 8 * hh;
      const int col = ct * 32 + l31;
      float xn[8];
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) xn[kk] = xrow[kk];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 wh = *reinterpret_cast<const bf16x8*>(wrow + 16 * ks);
        const bf16x8 wm = *reinterpret_cast<const bf16x8*>(wrow + PL + 16 * ks);
        const bf16x8 wl = *reinterpret_cast<const bf16x8*>(wrow + 2 * PL + 16 * ks);
        bf16x8 xh, xm, xl;
        nm_split8(xn, xh, xm, xl);
        if (ks + 1 < 4) {          // the next step's inputs are read under this step's products
#pragma unroll
          for (int kk = 0; kk < 8; ++kk) xn[kk] = xrow[16 * (ks + 1) + kk];
        }
        acc = NM_MFMA(xl, wh, acc);
        acc = NM_MFMA(xh, wl, acc);
        acc = NM_MFMA(xm, wm, acc);
        acc = NM_MFMA(xm, wh, acc);
        acc = NM_MFMA(xh, wm, acc);
        acc = NM_MFMA(xh, wh, acc);
      }
      // the two halves of the sum meet through h1's place: wavefront kh finishes accumulator registers 8 kh .. 8 kh + 7
      // (rows 16 kh + 4 hh + {0 .. 3, 8 .. 11} of its tile) and hands the partner its sums of the other eight
      if (kh == 0) {
#pragma unroll
        for (int r = 8; r < 16; ++r) H1[(rt * 32 + NM_ROW(r, hh)) * NM_LD + col] = acc[r];
      } else {
#pragma unroll
        for (int r = 0; r < 8; ++r) H1[(rt * 32 + NM_ROW(r, hh)) * NM_LD + col] = acc[r];
      }
      __syncthreads();
      const float bv = B1[col];
      auto finish = [&](auto half) {
        constexpr int KH = decltype(half)::value;
        uint16_t gate[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int row = rt * 32 + NM_ROW(8 * KH + r, hh);
          const float other = H1[row * NM_LD + col];
          float v = (KH ? other + acc[8 + r] : acc[r] + other) + bv;      // inputs 0 .. 63, then 64 .. 127, whichever wavefront adds
          v = v > 0.f ? v : 0.f;
          if (drop) {
            int rowv = row;
            asm volatile("" : "+v"(rowv));    // (formed here: 64-bit counters kept across the tile loop would cost registers all the way)
            const uint32_t draw = (uint32_t)(nm_mix(seed, (uint64_t)(p0 + rowv) * 64ull + (uint64_t)col) >> 32);
            v = draw >= thr ? v * inv_keep : 0.f;
          }
          v = row < PT ? v : 0.f;
          H1[row * NM_LD + col] = v;
          gate[r] = v > 0.f ? (uint16_t)0x3F80 : (uint16_t)0;       // bf16 1.0 / 0.0
          G16[row * NM_LG + col] = gate[r];
        }
#pragma unroll
        for (int q = 0; q < 2; ++q)     // rows 8 (2 KH + q) + 4 hh .. + 3 of the tile sit in registers 4 q .. 4 q + 3 of this half
          *reinterpret_cast<uint2*>(GT16 + col * NM_LG + rt * 32 + 8 * (2 * KH + q) + 4 * hh) =
              make_uint2((uint32_t)gate[4 * q] | ((uint32_t)gate[4 * q + 1] << 16), (uint32_t)gate[4 * q + 2] | ((uint32_t)gate[4 * q + 3] << 16));
      };
      if (kh == 0) finish(std::integral_constant<int, 0>{});
      else finish(std::integral_constant<int, 1>{});
    }
    __syncthreads();
    NM_STAMP(2);
    // ---- 3. logit = c0 + tg . zg + t2 . h1 (TP threads per pair, left in ZD), BCE term (fp64), dL/dlogit
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
      static_assert(TP == 8, "lanes per pair");
      x = nm_sum8(x);
      if (tid % TP == 0) ZD[p] = x + (c0 + ZD[p]);
    }
    __syncthreads();
    // the fp64 part on ONE wavefront with every lane at work (pair = lane): spread over the eight, each would
    // issue the whole sequence for eight active lanes
    if (wave == 0) {
      int p = lane;
      asm volatile("" : "+v"(p));
      float dl = 0.f;
      if (p < PT) {
        const float x = ZD[p];
        if (a.logits) a.logits[p0 + p] = x;
        const double xd = (double)x, y = (double)yraw;
        const NmBce t = nm_bce_term(xd, y);
        lsum += t.term;
        dl = (float)((t.sig - y) * a.inv_m);
      }
      DL[p] = dl;
      sdl += dl;
    }
    __syncthreads();
    NM_STAMP(3);
    // ---- 4. v += dl h1, db1 += gate dl (t2 / keep comes in at the end); vg += dl zg, item GMF gradients out
    {
      const int c = tid & 63;
#pragma unroll
      for (int p = wave * PW; p < wave * PW + PW; ++p) {
        const float h = H1[p * NM_LD + c], dl = DL[p];
        va += dl * h;
        db1a += h > 0.f ? dl : 0.f;
      }
    }
    float4 gu[KI];      // the pair's share of the user's GMF gradient: into LDS once every wavefront is done with h1
#pragma unroll
    for (int k = 0; k < KI; ++k) {
      const int p = qp + QS * k;
      const float dl = DL[p];
      vga[0] += dl * (rug[k].x * rig[k].x); vga[1] += dl * (rug[k].y * rig[k].y);
      vga[2] += dl * (rug[k].z * rig[k].z); vga[3] += dl * (rug[k].w * rig[k].w);
      const float4 dz = make_float4(dl * tg4.x, dl * tg4.y, dl * tg4.z, dl * tg4.w);
      if (p < PT) *reinterpret_cast<float4*>(a.di_g + (p0 + p) * NM_D + qc) = make_float4(dz.x * rug[k].x, dz.y * rug[k].y, dz.z * rug[k].z, dz.w * rug[k].w);
      gu[k] = make_float4(dz.x * rig[k].x, dz.y * rig[k].y, dz.z * rig[k].z, dz.w * rig[k].w);
    }
    // the next tile's rows travel while the gradient products below run
    nm_load_rows(a, ou, oi, qc, rum, rim, rug, rig);
    __syncthreads();
    NM_STAMP(4);
#pragma unroll
    for (int k = 0; k < KI; ++k) {
      float* gr = GU + (qp + QS * k) * NM_LD + qc;
      gr[0] = gu[k].x; gr[1] = gu[k].y; gr[2] = gu[k].z; gr[3] = gu[k].w;
    }
    // ---- 5. The gate is 0 / 1, so dz1 = gate (dl t2 / keep) factors: dW1 = (t2 / keep) . gate^T (dl x) and
    //         dx = dl . gate (t2 / keep . W1).  With the fp32 factor cut into three bf16 planes, a product with the
    //         0 / 1 operand on v_mfma_f32_32x32x16_bf16 accumulates exactly the fp32 terms.
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
          dW1a[ht] = NM_MFMA(ga, bh, dW1a[ht]);
          dW1a[ht] = NM_MFMA(ga, bm, dW1a[ht]);
          dW1a[ht] = NM_MFMA(ga, bl, dW1a[ht]);
        }
      }
    }
    {
      f32x16 acc1;
      const int rt = wave & 1, cq = wave >> 1;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 ga = *reinterpret_cast<const bf16x8*>(G16 + (rt * 32 + l31) * NM_LG + 16 * ks + 8 * hh);
        const bf16x8 wh = *reinterpret_cast<const bf16x8*>(WH16 + (cq * 32 + l31) * NM_LG + 16 * ks + 8 * hh);
        const bf16x8 wm = *reinterpret_cast<const bf16x8*>(WM16 + (cq * 32 + l31) * NM_LG + 16 * ks + 8 * hh);
        acc1 = NM_MFMA(ga, wlo[ks], acc1);
        acc1 = NM_MFMA(ga, wm, acc1);
        acc1 = NM_MFMA(ga, wh, acc1);
      }
      NM_STAMP(5);
      __syncthreads();    // every read of x is done: it now takes dx
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rt * 32 + NM_ROW(r, hh);
        X[row * NM_LDX + cq * 32 + l31] = acc1[r] * DL[row];
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
    printf("block %d phases (10 ns ticks): wait %lld  stage %lld  h1 %lld  logit %lld  dz %lld  mfma %lld  dx %lld  out %lld  | tile %lld\n", (int)blockIdx.x,
           stamp[8] - stamp[0], stamp[1] - stamp[0], stamp[2] - stamp[1], stamp[3] - stamp[2], stamp[4] - stamp[3], stamp[5] - stamp[4], stamp[6] - stamp[5],
           stamp[7] - stamp[6], stamp[7] - stamp[0]);
#endif

  if (bad && a.err_flag) atomicOr(a.err_flag, RX_ERR_INDEX);
  // ---- this workgroup's partial: [dW1 | db1 | v | vg | sum dl]; the thread-level pieces meet in LDS, fixed order
  float* wp = a.raw + (size_t)blockIdx.x * NM_RAW;
  {
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
  }
  float* red = X;        // [db1: WAVES x 64 | v: WAVES x 64 at 512 | vg: QS x 64 at 1024 | sum dl: 64 at 3072]
  red[wave * 64 + lane] = db1a;
  red[512 + wave * 64 + lane] = va;
#pragma unroll
  for (int e = 0; e < 4; ++e) red[1024 + qp * 64 + qc + e] = vga[e];
  if (wave == 0) red[3072 + lane] = sdl;
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
  static bool attr_set[64] = {};
  if (device >= 64 || !attr_set[device]) {   // LDS beyond 64 KiB per workgroup has to be asked for, per device
    RX_HIP(hipFuncSetAttribute((const void*)neumf_step_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    RX_HIP(hipFuncSetAttribute((const void*)neumf_step_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (device < 64) attr_set[device] = true;
  }
  hipStream_t st = (hipStream_t)stream;
  if (r->labels_f64) hipLaunchKernelGGL(neumf_step_kernel<true>, dim3(grid), dim3(NM_THREADS), lds, st, a);
  else hipLaunchKernelGGL(neumf_step_kernel<false>, dim3(grid), dim3(NM_THREADS), lds, st, a);
  RX_LAUNCH_CHECK();
  NmFin f{};
  f.raw = a.raw; f.n_parts = grid; f.wgrad = r->wgrad;
  f.W2 = r->w2; f.b2 = r->b2; f.Wg = r->wg; f.bg = r->bg; f.wf = r->wf;
  f.loss_part = a.loss_part; f.count = (double)r->B * (double)r->N; f.loss = r->loss; f.seed = const_cast<int64_t*>(r->seed);
  hipLaunchKernelGGL(neumf_finish_kernel, dim3(NM_FIN_BLOCKS), dim3(NM_FB), 0, st, f);
  RX_LAUNCH_CHECK();
  return 0;
}
