/*
 * recsys_hip.h — C ABI of librecsys_hip.so (MI355X / gfx950).
 *
 * The reference (huyqd/recsys) exposes its embedding + interaction hot path only as
 * PyTorch nn.Modules that issue stock torch ops; it has no FFI seam.  This header is the
 * seam the drop-in modules in recsys_amd/ call instead of those torch ops.  Each entry
 * point cites the reference op sequence it replaces (paths relative to the reference
 * checkout).
 *
 * Conventions
 *  - Every function returns 0 on success, nonzero on failure; rx_last_error() then
 *    returns a thread-local message.  Arguments are validated on the host before any
 *    launch (null pointers, sizes, supported widths).
 *  - All pointers named d_* / tables / outputs are DEVICE pointers owned by the caller
 *    (the host side passes tensor.data_ptr()).  The library never allocates, frees or
 *    keeps them.  Scratch memory is passed in as `ws` with `ws_bytes`; its size comes
 *    from the matching *_workspace_bytes() query.
 *  - `device` is the HIP device ordinal, `stream` a hipStream_t (as void*).  Calls are
 *    asynchronous on that stream, never synchronise it, and are re-entrant.
 *  - Embedding tables are row-major fp32 [rows, dim]; indices are int64 (the dtype the
 *    reference's batches carry, recsys/dataset.py:255-262).
 *  - `err_flag` (device int32*, may be NULL): kernels never dereference an out-of-range
 *    row and OR a bit into *err_flag (RX_ERR_INDEX) when they meet one.  The host side checks
 *    the word and raises IndexError, as torch does for nn.Embedding.  What happens to the
 *    occurrence itself: gathers and forward kernels leave it out (zeros); the index
 *    preparation (rx_sort_dedup*, rx_fg_group) clamps it into the table / into its field, so a
 *    backward pass run on a flagged batch adds that occurrence's gradient to the clamped row.
 *    Nothing is ever written outside the rows the batch names plus such clamped rows; beyond
 *    that the tables of a flagged step are unspecified, and the step should be treated as failed.
 */
#ifndef RECSYS_HIP_H
#define RECSYS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RX_ABI_VERSION 1
#define RX_ERR_INDEX 1    /* bit set in *err_flag when an index was out of range */
#define RX_ERR_CAPACITY 2 /* bit set when a fixed-capacity exchange buffer was too small */

#define RX_MAX_SEGS 8   /* segments per rx_gather_concat call */
#define RX_MAX_DIM 1024 /* widest embedding row the fused kernels take */

int rx_abi_version(void);
const char* rx_last_error(void);
/* Per-device set-up, no counterpart in the reference: makes `device` current and resolves the
 * addresses of the library's own device arrays there (rx_bce_fwd_noalloc's scratch).  Optional for
 * eager use -- the first eager call that needs an address resolves it -- but REQUIRED before
 * rx_bce_fwd_noalloc is recorded into a HIP graph on that device: the lookup may load the code
 * object (an allocation) and is not legal under stream capture.  Idempotent, thread-safe; call it
 * outside any capture. */
int rx_init(int device);
/* Measurement aid without a counterpart in the reference: writes the device wall clock (ticks of
 * the constant 100 MHz counter, 10 ns each) to *d_slot in stream order.  Works between the nodes
 * of a captured HIP graph, where HIP events cannot be queried per replay. */
int rx_wall_clock(int64_t* d_slot, int device, void* stream);

/* ------------------------------------------------------------------ gather / assemble
 * Replaces nn.Embedding lookups followed by unsqueeze/repeat/cat/view when a tower input
 * is built: recsys/models/neuralcf/mlp.py:67-76, neumf.py:92-101, widendeep.py:79-89,
 * deep_factorization_machine.py:94-102.  Rows are copied bit-exactly.
 *
 * out is [out_rows, out_ld] fp32.  For segment s and output row i:
 *   j   = (i / s.div) * s.stride + s.k          element of s.idx that row i uses
 *   row = s.idx[j] + s.base                     row of s.table (base = field offset)
 *   out[i, s.col : s.col + s.dim] = s.table[row, :]
 * div = N broadcasts one gathered row over N consecutive output rows (user rows over the
 * N items of a sample); stride/k pick field k out of a [B, stride] index matrix.
 * bag > 1 sum-pools: the block receives the sum, in bag order, of the rows named by index
 * elements j .. j + bag - 1 (k + bag <= stride) -- the .sum(1) over a field block that follows a
 * multi-field lookup (recsys/models/matrix_factorization/factorization_machine.py:40-46).
 * All segments of a call are served by ONE launch when every dim and col is a multiple of 4,
 * out_ld too, pointers 16-byte aligned and the bags of a row total at most 32 index entries;
 * other shapes take one launch per segment.
 */
typedef struct {
  const float* table;   /* [rows, dim] */
  const int64_t* idx;   /* int64 indices */
  int64_t rows;         /* rows in table (bounds check) */
  int64_t base;         /* added to every index */
  int32_t dim;          /* row width */
  int32_t col;          /* first output column */
  int32_t div;          /* >= 1 */
  int32_t stride;       /* >= 1 */
  int32_t k;            /* 0 <= k < stride */
  int32_t bag;          /* 0 or 1: one row; > 1: sum of `bag` consecutive index elements' rows */
} rx_gather_seg;

int rx_gather_concat(const rx_gather_seg* segs, int n_segs, float* d_out, int64_t out_rows,
                     int64_t out_ld, int32_t* err_flag, int device, void* stream);

/* Backward companion for one segment: the gradient of the q-th gathered row
 * (q = 0 .. out_rows/div - 1, the row that used index element q*stride + k) is
 *   occ[q, :] = sum_{r < div} d_out[q*div + r, col : col+dim]
 * occ is [out_rows/div, dim] fp32. */
int rx_gather_concat_bwd(const float* d_dout, int64_t out_rows, int64_t out_ld, int32_t col,
                         int32_t dim, int32_t div, float* d_occ, int device, void* stream);

/* ------------------------------------------------------------------ MF-family score
 * Replaces, fused, without the [B,N,D] temporaries:
 *   VanillaMF      recsys/models/matrix_factorization/vanilla_mf.py:29-31
 *   BiasMF         bias_mf.py:16-18,28-35
 *   SideFeaturesMF side_features_mf.py:24-40
 *   TemporalMF     temporal_mf.py:34-54
 *   GMF            recsys/models/neuralcf/gmf.py:49-57 (lin_w != NULL: Linear(D,1) folded in)
 *
 * logits[b,n] = sum_d (U[u_b,d] + O[o_b,d]) * V[i_bn,d] * lin_w[d]  + lin_b
 *             + bias + user_bias[u_b] + item_bias[i_bn]
 *             + sum_d UT[u_b,d] * T[t_b,d]
 * Any of O, lin_w/lin_b, the three biases and UT/T may be NULL (term absent).
 */
typedef struct {
  const float* user_table;  int64_t n_users;      /* U  [n_users, dim] */
  const float* item_table;  int64_t n_items;      /* V  [n_items, dim] */
  const float* occ_table;   int64_t n_occ;        /* O  [n_occ, dim] or NULL */
  const float* utemp_table;                       /* UT [n_users, dim] or NULL */
  const float* temp_table;  int64_t n_temp;       /* T  [n_temp, dim] or NULL */
  const float* user_bias;                         /* [n_users] or NULL */
  const float* item_bias;                         /* [n_items] or NULL */
  const float* bias;                              /* [1] or NULL */
  const float* lin_w;                             /* [dim] or NULL */
  const float* lin_b;                             /* [1] or NULL */
  const int64_t* users;                           /* [B] */
  const int64_t* items;                           /* [B, N] */
  const int64_t* occs;                            /* [B] or NULL */
  const int64_t* temps;                           /* [B] or NULL */
  int64_t B; int32_t N; int32_t dim;
  /* rx_mf_score_bwd only, all 0 otherwise: the final sum that a rx_mf_score_fwd_bce call with d_loss = NULL left
   * out.  loss_out[0] = sum(loss_parts[0 .. n_loss_parts)) / loss_count, added in the order of that call's own
   * finishing launch (bit for bit the same loss), by the first block of the backward launch. */
  const double* loss_parts; int32_t n_loss_parts; int32_t pad_; int64_t loss_count; double* loss_out;
} rx_mf_args;

int rx_mf_score_fwd(const rx_mf_args* a, float* d_logits, int32_t* err_flag, int device,
                    void* stream);

/* rx_mf_score_fwd followed by the loss of the MF family and GMF (vanilla_mf.py:35-39, gmf.py:41-45:
 * F.binary_cross_entropy_with_logits, mean over B * N, fp64) in one launch plus the one-block sum of the
 * block partials: d_loss [1] fp64, d_dlogits [B, N] = (sigmoid(x) - y) / (B N) (what the backward starts
 * from), d_logits [B, N] (may be NULL).  ws: rx_mf_score_fwd_bce_workspace_bytes bytes.
 * d_loss may be NULL: the block partials then stay in ws (rx_mf_score_fwd_bce_partials(B, dim) of them) and the
 * backward launch adds them up (rx_mf_args.loss_parts = ws, loss_count = B * N): one launch less per train step. */
size_t rx_mf_score_fwd_bce_workspace_bytes(int64_t B, int32_t dim);
int32_t rx_mf_score_fwd_bce_partials(int64_t B, int32_t dim);
int rx_mf_score_fwd_bce(const rx_mf_args* a, const void* d_labels, int labels_f64, float* d_logits,
                        float* d_dlogits, double* d_loss, void* ws, size_t ws_bytes, int32_t* err_flag,
                        int device, void* stream);

/* Backward of the above given g = dL/dlogits [B,N].  Writes per-occurrence gradients:
 *   d_user_occ [B, dim]   = sum_n g_bn * V[i_bn] * lin_w        (also the gradient of O rows)
 *   d_item_occ [B*N, dim] = g_bn * (U+O) * lin_w
 *   d_utemp_occ[B, dim]   = (sum_n g_bn) * T[t_b]      d_temp_occ [B, dim] = (sum_n g_bn) * UT[u_b]
 *   d_gsum     [B]        = sum_n g_bn                 (gradient of user_bias[u_b])
 *   d_linw_part[grid, dim]  per-block partial sums of g_bn*(U+O)*V  (reduced in block order
 *                           by rx_colsum_partials; NULL when lin_w is NULL)
 * rx_mf_bwd_grid() returns the number of blocks = rows of d_linw_part.  */
int64_t rx_mf_bwd_grid(int64_t B, int32_t dim);
int rx_mf_score_bwd(const rx_mf_args* a, const float* d_g, float* d_user_occ, float* d_item_occ,
                    float* d_utemp_occ, float* d_temp_occ, float* d_gsum, float* d_linw_part,
                    int device, void* stream);

/* out[c] = sum over r in block order of part[r, c]; fixed order, bitwise reproducible. */
int rx_colsum_partials(const float* d_part, int64_t rows, int32_t cols, float* d_out, int device,
                       void* stream);

/* ------------------------------------------------------------------ GMF product (NeuMF)
 * prod[b*N+n, col:col+dim] = U[u_b] * V[i_bn]   — recsys/models/neuralcf/neumf.py:84-89 ahead
 * of linear_gmf.  Backward: d_user_occ[b] = sum_n dprod[b,n]*V[i_bn]; d_item_occ[b,n] = dprod*U. */
int rx_gmf_mul_fwd(const float* user_table, int64_t n_users, const float* item_table,
                   int64_t n_items, const int64_t* users, const int64_t* items, int64_t B,
                   int32_t N, int32_t dim, float* d_out, int64_t out_ld, int32_t col,
                   int32_t* err_flag, int device, void* stream);
int rx_gmf_mul_bwd(const float* user_table, int64_t n_users, const float* item_table,
                   int64_t n_items, const int64_t* users, const int64_t* items, int64_t B,
                   int32_t N, int32_t dim, const float* d_dout, int64_t out_ld, int32_t col,
                   float* d_user_occ, float* d_item_occ, int device, void* stream);

/* ------------------------------------------------------------------ FM second order
 * Replaces recsys/models/matrix_factorization/factorization_machine.py:33-64 and the same
 * block of recsys/models/neuralcf/deep_factorization_machine.py:62-92 (one shared table,
 * field offsets, D=1 bias table gathered with the same indices, sum-square minus
 * square-sum).  ex_idx is [B, F] with per-field offsets ex_base[f] added on the fly;
 * item_idx is [B, N] with item_base added.
 *
 * logits[b,n] = sum_f bias[ex_f] + bias[item_n]
 *             + 0.5 * sum_d [ (sum_f e_f + v_n)^2 - (sum_f e_f^2 + v_n^2) ]_d
 * deep_out (may be NULL) receives the gathered rows as the DeepFM tower input
 * [B*N, deep_ld]: item row first, then the F rows (deep_factorization_machine.py:94-102).
 * d_logits may be NULL when only the gathered rows are wanted (DCN-v2 input).
 * With N == 1 and F = 25 this is the 26-field Criteo-shaped FM.
 */
typedef struct {
  const float* feat;        /* [n_feat, dim] */
  const float* bias;        /* [n_feat] (the [n_feat,1] table) or NULL */
  int64_t n_feat;
  const int64_t* ex_idx;    /* [B, F], row stride ex_ld */
  const int64_t* ex_base;   /* device [F] offsets */
  const int64_t* item_idx;  /* [B, N], row stride item_ld */
  int64_t item_base;
  int64_t ex_ld;            /* >= F: lets ex_idx / item_idx be column blocks of one [B, F+N] matrix */
  int64_t item_ld;          /* >= N */
  int64_t deep_ld;          /* row stride of deep_out / d_deep, >= (1+F)*dim (room for dense features) */
  int64_t B; int32_t F; int32_t N; int32_t dim;
  int32_t flags;            /* RX_FM_NO_SECOND_ORDER: bias sums only (the wide part of
                               recsys/models/neuralcf/widendeep.py:69-77); rows still feed deep_out */
  int64_t feat_ld;          /* row stride of feat in floats; 0 = dim (rows back to back) */
  int64_t bias_ld;          /* stride of bias in floats; 0 = 1.  A table whose rows are padded to whole
                               128-byte lines can hold a row's bias entry in the row's own line
                               (bias = feat + dim, bias_ld = feat_ld): the bias read then costs no
                               memory access of its own */
} rx_fm_args;
#define RX_FM_NO_SECOND_ORDER 1

/* d_tsum (may be NULL; N == 1 only) [B, dim] receives T_b = sum of all 1+F rows of sample b,
 * the only per-sample state rx_segreduce_fm needs for the backward. */
int rx_fm2_fwd(const rx_fm_args* a, float* d_logits, float* d_deep_out, float* d_tsum,
               int32_t* err_flag, int device, void* stream);

/* rx_fm2_fwd followed by the loss of recsys/models/matrix_factorization/factorization_machine.py:68-72
 * (F.binary_cross_entropy_with_logits, mean, fp64) in one launch, for N == 1: d_loss [1] fp64,
 * d_logits [B] (may be NULL), and per sample the record the field-grouped backward reads:
 * d_gt[b, 0:dim] = g_b * T_b, d_gt[b, dim] = g_b with g_b = (sigmoid(x_b) - y_b) / B (row stride ld_gt
 * >= dim + 1, a multiple of 4).  Two launches: the forward with per-block loss sums, and the
 * one-block sum of those.  ws: rx_fm2_fwd_bce_workspace_bytes bytes. */
size_t rx_fm2_fwd_bce_workspace_bytes(int64_t B, int32_t dim);
/* d_loss may be NULL: the final sum is then left to rx_fg_reduce (rx_fg_reduce_args.loss_parts = ws,
 * n_loss_parts = rx_fm2_fwd_bce_partials(B, dim), loss_count = B), which must follow on the same stream. */
int64_t rx_fm2_fwd_bce_partials(int64_t B, int32_t dim);
int rx_fm2_fwd_bce(const rx_fm_args* a, const void* d_labels, int labels_f64, float* d_logits,
                   float* d_gt, int64_t ld_gt, double* d_loss, void* ws, size_t ws_bytes,
                   int32_t* err_flag, int device, void* stream);

/* Backward given g [B,N] (and optionally d_deep [B*N,(1+F)*dim]):
 *   d_occ [(B*F + B*N), dim]: rows 0..B*F-1 = gradient of e_(b,f); then B*N item rows
 *   d_bias_occ [(B*F + B*N)]: gradient of the bias entries in the same order (NULL ok)
 * d e_f = G (S - e_f) + sum_n g_n v_n ; d v_n = g_n S ; G = sum_n g_n ; S = sum_f e_f. */
int rx_fm2_bwd(const rx_fm_args* a, const float* d_g, const float* d_deep, float* d_occ,
               float* d_bias_occ, int device, void* stream);

/* Flat index list in the order of d_occ above (row ids with offsets applied), int64. */
int rx_fm2_flat_indices(const rx_fm_args* a, int64_t* d_flat, int device, void* stream);

/* ------------------------------------------------------------------ BCE with logits
 * F.binary_cross_entropy_with_logits(logits f32, labels f64|f32), mean reduction — call
 * sites vanilla_mf.py:35-39, factorization_machine.py:68-72, gmf.py:41-45, mlp.py:59-63,
 * neumf.py:75-79, widendeep.py:96-100, deep_factorization_machine.py:109-113.  The loss is
 * accumulated and returned in fp64 (labels are fp64 in the reference, dataset.py:184).
 * d_dlogits (may be NULL) = (sigmoid(x) - y) / M.  ws: rx_bce_workspace_bytes(M). */
size_t rx_bce_workspace_bytes(int64_t M);
int rx_bce_fwd(const float* d_logits, const void* d_labels, int labels_f64, int64_t M,
               double* d_loss, float* d_dlogits, void* ws, size_t ws_bytes, int device,
               void* stream);

/* rx_bce_fwd for callers that cannot allocate (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  It has no counterpart in the reference and none of its own in
 * the op list above: same inputs, outputs and results as rx_bce_fwd, but the per-block partial sums
 * live in an 8 KiB device array inside the library (one copy per GPU, held until the process ends)
 * instead of a caller workspace, and the call issues no hipMalloc / hipFree.  M is limited to
 * 2^28 logits; a larger M is rejected with an error.  All calls on one GPU share that array: calls
 * that overlap on one GPU (different streams) race and must be ordered by the caller.  The array's
 * address on a device is resolved by rx_init(device) or, failing that, by the first EAGER call on
 * that device; a first call made while `stream` is being captured is refused with an error (so the
 * steady-state call, captured or not, touches no allocator: none of the library's, the runtime's or
 * the caller's). */
int rx_bce_fwd_noalloc(const float* d_logits, const void* d_labels, int labels_f64, int64_t M,
                       double* d_loss, float* d_dlogits, int device, void* stream);

/* ------------------------------------------------------------------ NeuMF train step in one launch
 * recsys/models/neuralcf/neumf.py:68-106 (list-wise forward: user / item rows of the GMF and MLP tables,
 * GMF product -> linear_gmf, cat([user, item]) -> linear_mlp (Linear, ReLU, Dropout, Linear), cat([gmf, mlp])
 * -> linear_final), its loss neumf.py:75-79 (mean BCE with logits over the B x N pairs, fp64) and what
 * loss.backward() (recsys/utils.py:98) computes for it, for embedding_dim 64 and the default tower
 * 2D -> D -> D/2: one kernel gathers, runs forward and backward of every pair on the matrix cores (exact
 * fp32 products) and writes
 *   du_gmf, du_mlp [B, 64]      gradient of each sample's user rows (summed over its N pairs, pair order)
 *   di_gmf, di_mlp [B * N, 64]  gradient of each pair's item rows
 *   wgrad [rx_neumf_step_weight_floats()]  gradients of linear_mlp.0.weight [64,128] | .bias [64] |
 *                               linear_mlp.3.weight [32,64] | .bias [32] | linear_gmf.weight [32,64] | .bias [32] |
 *                               linear_final.weight [64] | .bias [1], back to back
 *   loss (fp64 mean), logits [B * N] (optional)
 * The per-row embedding gradients then go through rx_sort_dedup + rx_segreduce like any other lookup.
 * p_drop: the Dropout between the tower's layers (0 in eval mode); its keep mask is a counter-based draw
 * keyed by (*seed, pair, unit), and the call advances *seed by one (seed may be NULL: stream 0, not advanced).
 * Order of additions is fixed by (B, N): results are bitwise reproducible.  ws: rx_neumf_step_workspace_bytes,
 * 256-byte aligned. */
typedef struct {
  const float* user_gmf; const float* item_gmf; const float* user_mlp; const float* item_mlp;   /* [n_users | n_items, 64] */
  int64_t n_users, n_items;
  const int64_t* users;    /* [B] */
  const int64_t* items;    /* [B, N] */
  const void* labels;      /* [B, N] f64 (labels_f64 != 0) or f32 */
  int32_t labels_f64, N;
  int64_t B;
  int32_t dim, pad_;
  const float* w1; const float* b1; const float* w2; const float* b2;     /* linear_mlp.0, linear_mlp.3 */
  const float* wg; const float* bg; const float* wf; const float* bf;     /* linear_gmf, linear_final */
  float p_drop; float pad2_;
  const int64_t* seed;     /* device counter of the dropout stream, or NULL */
  float* du_gmf; float* du_mlp; float* di_gmf; float* di_mlp;
  float* wgrad; double* loss; float* logits;
} rx_neumf_args;
int rx_neumf_step_supported(int32_t dim, int32_t N);
int64_t rx_neumf_step_weight_floats(void);
size_t rx_neumf_step_workspace_bytes(int64_t B, int32_t N);
int rx_neumf_step(const rx_neumf_args* a, void* ws, size_t ws_bytes, int32_t* err_flag, int device, void* stream);

/* ------------------------------------------------------------------ index sort / dedup
 * First stage of the deterministic backward that replaces aten embedding_dense_backward /
 * index_put_(accumulate=True) (what autograd runs for every nn.Embedding and 1-D bias
 * parameter of the reference models).  Stable LSD radix sort of (row, position) pairs,
 * then run-length compaction:
 *   sorted_pos [n] int32   positions 0..n-1 ordered by (row, position)
 *   sorted_slot[n] int32   index into uniq_rows of the row at each sorted position (may be NULL;
 *                          rx_segreduce needs it)
 *   uniq_rows  [n] int64   distinct rows ascending (first *n_uniq entries valid)
 *   seg_off    [n+1] int32 start of each row's run in sorted_pos; seg_off[*n_uniq] = n
 *   n_uniq     [1] int32   (device)
 *   inverse    [n] int32   slot of each occurrence's row in uniq_rows (may be NULL)
 * Rows must lie in [0, n_rows); n_rows <= 2^32.  Out-of-range rows set err_flag and are
 * clamped.  Equal to torch.sort(stable=True) / torch.unique(sorted=True, return_inverse). */
size_t rx_sort_dedup_workspace_bytes(int64_t n);
int rx_sort_dedup(const int64_t* d_idx, int64_t n, int64_t n_rows, int32_t* d_sorted_pos,
                  int32_t* d_sorted_slot, int64_t* d_uniq_rows, int32_t* d_seg_off, int32_t* d_n_uniq,
                  int32_t* d_inverse, void* ws, size_t ws_bytes, int32_t* err_flag, int device,
                  void* stream);
/* The same over the row list of an FM batch, read straight from the index matrices in `a`
 * (ex_idx, ex_base, item_idx, item_base, B, F, N, n_feat = rows of the table; the table pointers
 * are not touched): entry i < B*F is extra field i % F of sample i / F plus that field's base row,
 * the remaining B*N entries are the item rows plus item_base, i.e. exactly the list
 * rx_fm2_flat_indices writes, without writing it.  n = B*(F+N) sizes the outputs and the workspace. */
int rx_sort_dedup_fm(const rx_fm_args* a, int32_t* d_sorted_pos, int32_t* d_sorted_slot,
                     int64_t* d_uniq_rows, int32_t* d_seg_off, int32_t* d_n_uniq, int32_t* d_inverse,
                     void* ws, size_t ws_bytes, int32_t* err_flag, int device, void* stream);

/* ------------------------------------------------------------------ segmented reduce + update
 * Second stage.  For each distinct row u the gradient
 *   G_u = sum_{p in run u} occ[sorted_pos[p], :]
 * is formed without atomics in an order fixed by the sort (runs are cut every 32 sorted
 * positions: position order inside a piece, piece order across pieces, so hot rows cost no
 * more wall time than cold ones), then consumed according to `mode`:
 *   RX_SEG_DENSE   grad[row, :]  = G_u                   (grad pre-zeroed by the caller)
 *   RX_SEG_COMPACT grad[u, :]    = G_u
 *   RX_SEG_SGD     w[row] -= lr * scale * G_u                              (torch.optim.SGD)
 *   RX_SEG_ADAGRAD state[row] += (scale G_u)^2 ; w[row] -= lr * scale G_u / (sqrt(state)+eps)
 *                                                                      (torch.optim.Adagrad)
 *   RX_SEG_ROWWISE_ADAGRAD  state[row] (one float per row) += mean_d (scale G_u)^2
 * A companion scalar table addressed by the same rows (the [n,1] bias / wide table that the
 * reference gathers with the same indices, factorization_machine.py:40-51) can ride along:
 * d_occ2 [n] per-occurrence gradients, d_target2 / d_state2 its table (or dense / compact
 * gradient) and Adagrad state; all three NULL when absent.
 *   RX_SEG_SGD_SPEC (rx_segreduce, rx_segreduce_fm, rx_row_update; not the _multi calls): the clipped SGD step
 *                  in one pass when the clip does not bite -- the reduce takes the step with a factor of 1 and
 *                  leaves undo records where a compact gradient would go: d_state [n, 2 * dim] = G_u | row
 *                  before the step, d_state2 [n, 2] = companion gradient | entry before the step.
 *                  rx_row_update(mode = RX_SEG_SGD_SPEC, d_compact = those records, d_compact2 = the
 *                  companion's) ends at once for a factor of exactly 1 and otherwise redoes every row as
 *                  old - lr * (factor * G_u), bit for bit what RX_SEG_COMPACT + rx_row_update(RX_SEG_SGD) leave;
 *                  RX_SEG_SGD_SPEC_REDO redoes them whatever the factor (0 puts the rows back).
 * scale = *d_scale if d_scale != NULL else 1 (global-norm clip factor, recsys/utils.py:101).
 * d_sq_part (may be NULL) [rx_segreduce_grid(n, dim)] receives per-block sums of G_u^2.
 * ws: rx_segreduce_workspace_bytes(n, dim) bytes, 16-byte aligned (cross-piece partial sums). */
enum { RX_SEG_DENSE = 0, RX_SEG_COMPACT = 1, RX_SEG_SGD = 2, RX_SEG_ADAGRAD = 3,
       RX_SEG_ROWWISE_ADAGRAD = 4,
       RX_SEG_SGD_SPEC = 5,      /* rx_fg_reduce / rx_fg_row_update(_clip) only, see rx_fg_reduce_args.undo */
       RX_SEG_SGD_SPEC_REDO = 6, /* rx_fg_row_update(_clip) only: redo the rows from their records whatever the factor */
       /* The one-pass clipped SGD step with records of half the size, for wide rows (rx_segreduce, rx_segreduce_fm):
        * `state` [n][dim] / `state2` [n] keep only the row / entry of before the step.  When the factor turns out
        * below 1 the step is redone from scratch by three calls that cost a launch each and nothing more while
        * the factor is exactly 1:
        *   rx_row_update(RX_SEG_RESTORE_IF_CLIPPED, d_compact = records)   rows and entries back in place
        *   rx_segreduce_fm / rx_segreduce(RX_SEG_COMPACT_IF_CLIPPED, d_scale = the factor)   compact gradients again
        *   rx_row_update(RX_SEG_SGD_IF_CLIPPED)                            the clipped step
        * RX_SEG_RESTORE puts the rows back whatever the factor (gradients dropped, or a forward pass before the
        * step).  The result equals RX_SEG_COMPACT + RX_SEG_SGD bit for bit either way. */
       RX_SEG_SGD_SPEC_OLD = 7, RX_SEG_COMPACT_IF_CLIPPED = 8, RX_SEG_SGD_IF_CLIPPED = 9, RX_SEG_RESTORE = 10,
       RX_SEG_RESTORE_IF_CLIPPED = 11 };
int64_t rx_segreduce_grid(int64_t n, int32_t dim);
size_t rx_segreduce_workspace_bytes(int64_t n, int32_t dim);
int rx_segreduce(const float* d_occ, int64_t occ_ld, int64_t n, int32_t dim,
                 const int32_t* d_sorted_pos, const int32_t* d_sorted_slot,
                 const int64_t* d_uniq_rows, const int32_t* d_seg_off, const int32_t* d_n_uniq,
                 int mode, float* d_target, float* d_state, const float* d_occ2, float* d_target2,
                 float* d_state2, int64_t n_rows, float lr, float eps, const float* d_scale,
                 float* d_sq_part, void* ws, size_t ws_bytes, int device, void* stream);

/* Up to RX_SEG_MULTI_MAX tables in one pair of launches (the user / item tables of the MF and NCF
 * families, recsys/models/matrix_factorization/ and recsys/models/neuralcf/: a few thousand
 * occurrences per table, where a launch per table costs more than its work).  Every field as in
 * rx_segreduce; mode, lr, eps and the clip factor are shared.  Each table keeps its own blocks,
 * pieces and partials, so every result equals that of one rx_segreduce call per table bit for bit.
 * The widths of one call must share the lane layout (equal widths always do).  sq_part of table i
 * (may be NULL) has rx_segreduce_grid(n_i, dim_i) entries; ws holds the tables' workspaces back to
 * back: sum of rx_segreduce_workspace_bytes(n_i, dim_i).
 * rx_row_update_multi: rx_row_update for the same list, where occ / occ2 are the compact gradients
 * [*n_uniq, dim] / [*n_uniq], n the capacity of those buffers, target / target2 the tables; a
 * table may appear once per call. */
#define RX_SEG_MULTI_MAX 8
typedef struct rx_seg_table {
  const float* occ; int64_t occ_ld; int64_t n; int32_t dim;
  const int32_t* sorted_pos; const int32_t* sorted_slot; const int64_t* uniq_rows;
  const int32_t* seg_off; const int32_t* n_uniq;
  float* target; float* state;
  const float* occ2; float* target2; float* state2;
  int64_t n_rows;
  float* sq_part;
} rx_seg_table;
int rx_segreduce_multi(const rx_seg_table* tables, int32_t count, int mode, float lr, float eps,
                       const float* d_scale, void* ws, size_t ws_bytes, int device, void* stream);
int rx_row_update_multi(const rx_seg_table* tables, int32_t count, int mode, float lr, float eps,
                        const float* d_scale, int device, void* stream);

/* FM-fused variant for N == 1 (every sample contributes one row per field, e.g. the 26-field
 * Criteo-shaped FM): the gradient of row r in sample b is g_b * (T_b - e_r), so
 *   G_r = sum_b g_b T_b  -  (sum_b g_b) * e_r        and the bias gradient is sum_b g_b.
 * Nothing per-occurrence is materialised: the reduce walks the same sorted runs but reads
 * T [n_samples, dim] (row stride ld_T) and g [n_samples] (both cache resident), and subtracts
 * the row's own term when it reads the row for the update.  The occurrence list behind the
 * sort must be the rx_fm2_flat_indices order: n_samples*F entries (b, f) then n_samples item
 * entries.  d_table: the feature table (needed to read e_r when `mode` writes a gradient
 * buffer; ignored for the optimizer modes, which read d_target).  d_target2 / d_state2: the
 * bias table (or its dense / compact gradient) and its Adagrad state, or NULL.
 * d_deep (may be NULL) [n_samples, deep_ld]: gradient of the tower input that rx_fm2_fwd laid out
 * as [item row, field 0 .. F-1 rows, extra columns] (DeepFM / DCN-v2: the same gathered rows feed
 * the tower, deep_factorization_machine.py:94-103); occurrence (b, f) then also receives
 * d_deep[b, (1+f)*dim ...] and the item d_deep[b, 0 ...], so G_r gains the sum of those slices and
 * no per-occurrence gradient rows are written for the tower path either.  A model with a tower but
 * no FM term passes g = 0. */
int rx_segreduce_fm(const float* d_T, int64_t ld_T, const float* d_g, const float* d_deep,
                    int64_t deep_ld, int64_t n_samples, int32_t F, int32_t dim,
                    const int32_t* d_sorted_pos, const int32_t* d_sorted_slot,
                    const int64_t* d_uniq_rows, const int32_t* d_seg_off, const int32_t* d_n_uniq,
                    int mode, float* d_target, float* d_state, const float* d_table, float* d_target2,
                    float* d_state2, int64_t n_rows, float lr, float eps, const float* d_scale,
                    float* d_sq_part, void* ws, size_t ws_bytes, int device, void* stream);

/* ------------------------------------------------------------------ field-grouped backward (N == 1 field models)
 * The same two stages for the models whose samples name exactly one row in each of n_fields
 * DISJOINT, ascending row ranges of one table (the 26-field FM / DeepFM / DCN-v2 that generalise
 * recsys/models/matrix_factorization/factorization_machine.py:33-38 to one item per sample): what
 * autograd would do there is embedding_dense_backward over B * n_fields occurrences.
 *
 * rx_fg_group orders the occurrences without a full-width sort.  Position p = f * B + j holds the
 * j-th occurrence of field f in the order (row & 255, row >> 8, sample):
 *   srow [B * n_fields] uint32   table row of the occurrence (field base added)
 *   sb   [B * n_fields] uint32   its sample
 * Equal rows are adjacent, their samples ascend, a run never crosses a field.  The distinct rows of
 * the batch are the srow[p] with p the last position of a run (srow[p] != srow[p + 1]); as a set they
 * equal torch.unique of the flat row list.  Indices are field-local: an index outside
 * [0, field_base[f + 1] - field_base[f]) sets RX_ERR_INDEX and is clamped into its field (stricter
 * than nn.Embedding, which only checks the whole table: here the fields must not alias).
 * field_base holds HOST values.  Limits: B <= 65536, n_fields <= RX_FG_MAX_FIELDS, table rows
 * <= 2^32 - 16 (rx_fg_supported tells); fields of any size: the packed (row >> 8, sample) words are
 * 32 bits wide where bits(B - 1) + bits((largest field - 1) >> 8) <= 32 and 64 bits wide otherwise.
 * ex_idx [B, n_fields - 1] / item_idx [B, 1] with their row strides, as in rx_fm_args. */
#define RX_FG_MAX_FIELDS 64
typedef struct {
  const int64_t* ex_idx;   /* device [B, n_fields - 1], row stride ex_ld */
  const int64_t* item_idx; /* device [B, 1], row stride item_ld: the last field */
  int64_t ex_ld, item_ld;
  int64_t B;
  int32_t n_fields;
  int32_t pad_;
  int64_t field_base[RX_FG_MAX_FIELDS + 1]; /* host: first table row of field f; [n_fields] = rows of the table */
} rx_fg_args;
int rx_fg_supported(const rx_fg_args* a);
int rx_fg_dim_supported(int32_t dim); /* row widths rx_fg_reduce takes: 16, 32, 64, 128 */
size_t rx_fg_workspace_bytes(int64_t B, int32_t n_fields);
int rx_fg_group(const rx_fg_args* a, uint32_t* d_srow, uint32_t* d_sb, void* ws, size_t ws_bytes,
                int32_t* err_flag, int device, void* stream);

/* Reduce + consume over that order; meaning of the fields as in rx_segreduce_fm (T, g, optional
 * tower-input gradient `deep` laid out [last field | field 0 .. n_fields-2 | extra columns], modes,
 * companion [rows,1] table, clip factor, per-block sums of squares) with these differences:
 * RX_SEG_COMPACT writes the gradient of a run to target[p, :] / target2[p] with p the run's LAST
 * position (buffers of B * n_fields rows); sq_part has rx_fg_reduce_grid(n, dim) entries; g is read
 * as g[b * ld_g]; with premultiplied != 0, T[b, :] already holds g_b * T_b (what rx_fm2_fwd_bce
 * leaves).
 * One lane group per 8 consecutive positions, one segmented scan per wavefront, fixed tree:
 * bitwise reproducible. */
typedef struct {
  const float* T; int64_t ld_T;
  const float* g; int64_t ld_g;
  const float* deep; int64_t deep_ld;     /* NULL: no tower */
  int64_t B;
  int32_t n_fields, dim, mode, premultiplied;
  const uint32_t* srow; const uint32_t* sb;
  float* target; float* state; const float* table; float* target2; float* state2;
  int64_t n_rows;
  float lr, eps;
  const float* scale;                      /* NULL: 1 */
  float* sq_part;                          /* NULL: no sums of squares */
  int64_t ld_w, ld_w2;                     /* row strides (floats) of the TABLE and of the companion table:
                                              `table` in the gradient modes, `target` / `target2` in the update
                                              modes; 0 = dim / 1.  Gradient and state arrays are never strided.
                                              target2 == target + dim with ld_w2 == ld_w == 2 * dim (row and
                                              companion entry in one line, the rest of the line padding) makes
                                              the SGD update write whole lines: the padding is rewritten as 0 */
  /* RX_SEG_SGD_SPEC, the clipped SGD step (recsys/utils.py:101-103: clip_grad_norm_ then step) in ONE pass
   * over the distinct rows when the clip does not bite: the reduce takes the step with a clip factor of 1
   * (multiplying by 1 is exact, so these are the bits of the clipped step whenever norm <= max_norm), writes
   * sq_part, and leaves an undo record at each run's last position p:
   *   undo [p, 0 .. dim) = G_u, undo[p, dim .. 2 dim) = the row before the step        ([B * n_fields, 2 * dim])
   *   undo2[p, 0] = companion gradient, undo2[p, 1] = companion entry before the step  ([B * n_fields, 2])
   * In the line layout (above) undo2 is NULL and the pair sits in the row's own line, columns dim + 1 and
   * dim + 2.  rx_fg_row_update / rx_fg_row_update_clip with the same mode and d_compact = undo, d_compact2 =
   * undo2 then end at once when the clip factor is exactly 1, and otherwise redo every row from its record as
   * old - lr * (factor * G_u): bit for bit what RX_SEG_COMPACT + rx_fg_row_update(RX_SEG_SGD) leave.
   * RX_SEG_SGD_SPEC_REDO (update calls only) redoes the rows even for a factor of 1: with a factor of 0 it puts
   * the rows back as they were (a second backward pass before the step must read the old rows), and the step
   * then applies such records with the final factor. */
  float* undo; float* undo2;
  /* Deferred final sum of the BCE loss (may all be NULL / 0): rx_fm2_fwd_bce called with d_loss == NULL leaves
   * its per-block fp64 partial sums in its workspace (rx_fm2_fwd_bce_partials(B, dim) of them) instead of
   * launching the one-wavefront kernel that adds them; one wavefront of this launch then writes
   * loss_out[0] = sum(loss_parts) / loss_count with the same order of additions, bit for bit the same loss.
   * Takes a 5 us launch off the critical path between the forward pass and the reduce. */
  const double* loss_parts; int32_t n_loss_parts; int32_t pad2_; int64_t loss_count; double* loss_out;
} rx_fg_reduce_args;
int64_t rx_fg_reduce_grid(int64_t n, int32_t dim);
size_t rx_fg_reduce_workspace_bytes(int64_t n, int32_t dim);
int rx_fg_reduce(const rx_fg_reduce_args* a, void* ws, size_t ws_bytes, int device, void* stream);
/* Second phase of a clipped step: every run's row moves by its compact gradient (at the run's last
 * position) times *d_scale; mode SGD / ADAGRAD / ROWWISE_ADAGRAD; companion table in the same launch.
 * ld_w / ld_w2: row strides of d_w / d_w2 in floats (0 = dim / 1), as in rx_fg_reduce_args. */
int rx_fg_row_update(const float* d_compact, int32_t dim, const uint32_t* d_srow, int64_t n, int mode,
                     float* d_w, float* d_state, const float* d_compact2, float* d_w2, float* d_state2,
                     int64_t n_rows, float lr, float eps, const float* d_scale, int64_t ld_w, int64_t ld_w2,
                     int device, void* stream);
/* The same update with the global-norm clip factor (recsys/utils.py:101: clip_grad_norm_ over all
 * parameters) formed inside the launch: d_parts[0 .. n_parts) are ALL partial sums of squares of the
 * step (the sq_part of the one reduce that fed d_compact; nothing else may contribute), every block
 * adds them in the order rx_clip_scale_parts uses and scales by min(1, max_norm / (norm + 1e-6)).
 * d_scale_out[0] / d_norm_out[0] (may be NULL) receive the factor and the norm.  Saves the one-block
 * launch between reduce and update; bit-identical to rx_clip_scale_parts + rx_fg_row_update. */
int rx_fg_row_update_clip(const float* d_compact, int32_t dim, const uint32_t* d_srow, int64_t n, int mode,
                          float* d_w, float* d_state, const float* d_compact2, float* d_w2, float* d_state2,
                          int64_t n_rows, float lr, float eps, const float* d_parts, int64_t n_parts,
                          float max_norm, float* d_scale_out, float* d_norm_out, int64_t ld_w, int64_t ld_w2,
                          int device, void* stream);

/* Row update from a compact gradient [*n_uniq, dim] produced by RX_SEG_COMPACT (two-phase
 * step when a global clip factor must be known before any row moves). mode as above
 * (SGD / ADAGRAD / ROWWISE_ADAGRAD).  d_compact2 [*n_uniq] / d_w2 / d_state2: compact gradient,
 * table and Adagrad state of the companion scalar table updated in the same launch, or NULL. */
int rx_row_update(const float* d_compact, int32_t dim, const int64_t* d_uniq_rows,
                  const int32_t* d_n_uniq, int64_t n_max, int mode, float* d_w, float* d_state,
                  const float* d_compact2, float* d_w2, float* d_state2, int64_t n_rows, float lr,
                  float eps, const float* d_scale, int device, void* stream);

/* Dense full-table optimizers for MovieLens-sized tables and tower weights, one launch per
 * tensor: torch.optim.Adam(lr, betas, eps) exactly as recsys/pipelines/ build it
 * (e.g. recsys/pipelines/neuralcf/neumf.py:10), and SGD / Adagrad.  scale as above. */
int rx_dense_adam(float* d_w, const float* d_g, float* d_m, float* d_v, int64_t n, float lr,
                  float beta1, float beta2, float eps, int32_t step, const float* d_scale,
                  int device, void* stream);
/* The same step with the step count t read from device memory (d_step[0] >= 1; the caller advances
 * it on the stream before the call), so that a HIP graph recorded around the step does not freeze
 * t: torch.optim.Adam's ``capturable`` form of the step the pipelines build at
 * recsys/pipelines/neuralcf/neumf.py:10.  A counter below 1 leaves the weights untouched. */
int rx_dense_adam_dstep(float* d_w, const float* d_g, float* d_m, float* d_v, int64_t n, float lr,
                        float beta1, float beta2, float eps, const int32_t* d_step,
                        const float* d_scale, int device, void* stream);
int rx_dense_sgd(float* d_w, const float* d_g, int64_t n, float lr, const float* d_scale,
                 int device, void* stream);
int rx_dense_adagrad(float* d_w, const float* d_g, float* d_state, int64_t n, float lr, float eps,
                     const float* d_scale, int device, void* stream);

/* Sum of squares of a tensor (fp64 accumulate, fixed order) added into *d_acc (device
 * double); and scale = min(1, max_norm / (sqrt(*d_acc) + 1e-6)) as clip_grad_norm_ does
 * (recsys/utils.py:101). ws: rx_sumsq_workspace_bytes(n). */
size_t rx_sumsq_workspace_bytes(int64_t n);
int rx_sumsq_accumulate(const float* d_x, int64_t n, double* d_acc, void* ws, size_t ws_bytes,
                        int device, void* stream);
/* *d_acc += sum(x): adds the per-block partials that rx_segreduce wrote to d_sq_part. */
int rx_sum_accumulate(const float* d_x, int64_t n, double* d_acc, void* ws, size_t ws_bytes,
                      int device, void* stream);
int rx_clip_scale(const double* d_acc, float max_norm, float* d_scale, float* d_total_norm,
                  int device, void* stream);
/* The same factor from per-block partial sums of squares (the d_sq_part of rx_segreduce /
 * rx_segreduce_fm) in one launch: total = (d_acc ? *d_acc : 0) + sum(d_parts[0..n_parts)) +
 * sum(d_parts64[0..n_parts64)), the latter fp64 partials as rx_multi_sumsq writes them. */
int rx_clip_scale_parts(const float* d_parts, int64_t n_parts, const double* d_parts64,
                        int64_t n_parts64, const double* d_acc, float max_norm, float* d_scale,
                        float* d_total_norm, int device, void* stream);

/* Up to RX_MULTI_MAX small dense tensors (the towers' weights and biases) in one launch each:
 *   rx_multi_sumsq         d_partial[k * 32 + b] (fp64) = partial sums of squares of g[k]; the
 *                          buffer holds rx_multi_sumsq_partials(count) doubles and is consumed as
 *                          d_parts64 by rx_clip_scale_parts (clip_grad_norm_ over all parameters,
 *                          recsys/utils.py:101)
 *   rx_multi_dense_update  SGD (RX_SEG_SGD) or Adagrad (RX_SEG_ADAGRAD, state[k]) step of w[k] with
 *                          gradient g[k] * scale, as rx_dense_sgd / rx_dense_adagrad do per tensor */
#define RX_MULTI_MAX 24
typedef struct rx_tensor_list {
  float* w[RX_MULTI_MAX];
  const float* g[RX_MULTI_MAX];
  float* state[RX_MULTI_MAX];
  int64_t n[RX_MULTI_MAX];
  int32_t count;
} rx_tensor_list;
size_t rx_multi_sumsq_partials(int32_t count);
int rx_multi_sumsq(const rx_tensor_list* list, double* d_partial, int device, void* stream);
int rx_multi_dense_update(const rx_tensor_list* list, int mode, float lr, float eps,
                          const float* d_scale, int device, void* stream);
/* The tail of a clipped step (recsys/utils.py:101-103: clip_grad_norm_, optimizer.step) in ONE launch: the clip
 * factor from the step's partial sums of squares (rx_clip_scale_parts: fp32 parts of the table reduces, fp64 parts
 * of rx_multi_sumsq), the row updates of up to RX_SEG_MULTI_MAX tables from their compact gradients
 * (rx_row_update_multi) and the update of up to RX_MULTI_MAX dense tensors (rx_multi_dense_update: dense_mode SGD
 * or Adagrad, dense may be NULL).  Every block forms the factor itself, in the order of rx_clip_scale_parts, so the
 * results equal the three calls bit for bit; d_scale / d_total_norm (may be NULL) receive factor and norm. */
int rx_step_tail(const rx_seg_table* tables, int32_t count, int mode, float lr, float eps, const rx_tensor_list* dense,
                 int dense_mode, float dense_lr, float dense_eps, const float* d_parts, int64_t n_parts,
                 const double* d_parts64, int64_t n_parts64, float max_norm, float* d_scale, float* d_total_norm,
                 int device, void* stream);
/* torch.optim.Adam over every parameter of a model -- what the reference's pipelines build
 * (recsys/pipelines/neuralcf/neumf.py:10, matrix_factorization/vanilla_mf.py:10) -- as ONE launch for up to
 * RX_MULTI_MAX tensors: rx_dense_adam / rx_dense_adam_dstep element by element (d_step != NULL: the step count
 * is read from device memory), the same bits as one call per tensor. */
typedef struct rx_adam_list {
  float* w[RX_MULTI_MAX];
  const float* g[RX_MULTI_MAX];
  float* m[RX_MULTI_MAX];
  float* v[RX_MULTI_MAX];
  int64_t n[RX_MULTI_MAX];
  int32_t count;
} rx_adam_list;
int rx_multi_dense_adam(const rx_adam_list* list, float lr, float beta1, float beta2, float eps, int32_t step,
                        const int32_t* d_step, const float* d_scale, int device, void* stream);
/* out[0] = sum(x) (fp64 accumulate, fixed order): gradient of a scalar bias that is added to
 * every logit (bias_mf.py:12,17; the Linear(D,1) bias of gmf.py:20). Same workspace size. */
int rx_reduce_sum(const float* d_x, int64_t n, float* d_out, void* ws, size_t ws_bytes, int device,
                  void* stream);

/* ------------------------------------------------------------------ dense layers (MFMA)
 * nn.Linear (+ReLU) of the towers: recsys/models/neuralcf/mlp.py:27-36, neumf.py:21,36-47,
 * widendeep.py:10-21, deep_factorization_machine.py:10-21.  fp32 in/out; `bf16` != 0 rounds
 * the operands to bf16 for v_mfma_f32_32x32x16_bf16 with fp32 accumulation, otherwise the
 * exact-fp32 v_mfma_f32_32x32x2_f32 is used.
 *   rx_linear_fwd : Y[M,N] = act(X[M,K] W[N,K]^T + b)           act: 0 none, 1 ReLU
 *   rx_linear_bwd_x: dX[M,K] = dY[M,N] W[N,K]                    (accumulate != 0: dX += )
 *   rx_linear_bwd_w: dW[N,K] = dY^T X, db[N] = colsum(dY)        ws from the query below
 * ldx/ldy/ldd are row strides in elements (sub-blocks of wider matrices are allowed).
 * rx_relu_bwd: dZ = dY * (Y > 0) elementwise. */
int rx_linear_fwd(const float* X, int64_t ldx, const float* W, const float* b, float* Y,
                  int64_t ldy, int64_t M, int32_t N, int32_t K, int act, int bf16, int device,
                  void* stream);
int rx_linear_bwd_x(const float* dY, int64_t ldd, const float* W, float* dX, int64_t ldx,
                    int64_t M, int32_t N, int32_t K, int accumulate, int bf16, int device,
                    void* stream);
size_t rx_linear_bwd_w_workspace_bytes(int64_t M, int32_t N, int32_t K);
int rx_linear_bwd_w(const float* dY, int64_t ldd, const float* X, int64_t ldx, float* dW,
                    float* db, int64_t M, int32_t N, int32_t K, int bf16, void* ws,
                    size_t ws_bytes, int device, void* stream);
int rx_relu_bwd(const float* dY, const float* Y, float* dZ, int64_t n, int device, void* stream);
/* Backward of the pair ReLU + Dropout of the ReLU towers (recsys/models/neuralcf/mlp.py:27-36,
 * neumf.py:36-45: nn.Linear, nn.ReLU, nn.Dropout) from the pair's OUTPUT alone:
 * dZ = Y > 0 ? dY * scale : 0 with scale = 1 / (1 - p) -- Y is positive exactly where the ReLU was open
 * and the element was kept, so neither the keep mask nor the ReLU's own output has to be stored. */
int rx_relu_bwd_scaled(const float* dY, const float* Y, float* dZ, int64_t n, float scale, int device,
                       void* stream);

/* ---- BatchNorm1d of the reference's BN towers, fused with the ReLU and Dropout that follow it
 * (widendeep.py:6-24 and deep_factorization_machine.py:6-24: Linear, BatchNorm1d, ReLU,
 * Dropout per hidden layer).  x [M, C] row-major, statistics per column over the M rows.
 *   rx_bn_stats   mean[C], rstd[C] = 1/sqrt(biased var + eps); when the running buffers are given:
 *                 running = (1 - momentum) running + momentum (mean | unbiased var), as
 *                 torch.nn.BatchNorm1d does in training mode
 *   rx_bn_apply   out = keep * keep_scale * act(gamma (x - mean) rstd + beta); relu: act = ReLU;
 *                 d_keep (uint8 [M, C], may be NULL) is the dropout keep mask, keep_scale = 1/(1-p).
 *                 Evaluation mode passes the running mean and 1/sqrt(running var + eps).
 *   rx_bn_bwd     dz = gate ? dout * keep_scale * (out != 0) : dout;  dbeta = sum dz;
 *                 dgamma = sum dz xhat;  dx = gamma rstd (dz - dbeta/M - xhat dgamma/M) with batch
 *                 statistics, gamma rstd dz with fixed ones.  gate = relu or dropout was applied.
 * Sums are fp64 in a fixed order (slabs of rows, then slab order).  ws: rx_bn_workspace_bytes. */
size_t rx_bn_workspace_bytes(int64_t M, int32_t C);
int rx_bn_stats(const float* d_x, int64_t ldx, int64_t M, int32_t C, float eps, float momentum,
                float* d_mean, float* d_rstd, float* d_running_mean, float* d_running_var, void* ws,
                size_t ws_bytes, int device, void* stream);
int rx_bn_apply(const float* d_x, int64_t ldx, int64_t M, int32_t C, const float* d_mean,
                const float* d_rstd, const float* d_gamma, const float* d_beta, int relu,
                const uint8_t* d_keep, float keep_scale, float* d_out, int64_t ldo, int device,
                void* stream);
int rx_bn_bwd(const float* d_dout, int64_t ldd, const float* d_x, int64_t ldx, const float* d_out,
              int64_t ldo, int64_t M, int32_t C, const float* d_mean, const float* d_rstd,
              const float* d_gamma, int gate, float keep_scale, int batch_stats, float* d_dx,
              int64_t lddx, float* d_dgamma, float* d_dbeta, void* ws, size_t ws_bytes, int device,
              void* stream);

/* ------------------------------------------------------------------ DCN-v2 cross layer
 * No counterpart in the reference.  x_{l+1} = x0 * (x_l W^T + b) + x_l, W [d,d].
 * fwd: U = rx_linear_fwd(x_l, W, b) then xn = x0 * U + x_l; bwd elementwise part, given dXn:
 *   dU = dXn * x0 ; dX0 += dXn * U ; dXl = dXn.   All operands [rows, cols] with own row strides. */
int rx_cross_combine_fwd(const float* x0, int64_t ld0, const float* u, int64_t ldu, const float* xl,
                         int64_t ldl, float* xn, int64_t ldn, int64_t rows, int32_t cols, int device,
                         void* stream);
int rx_cross_combine_bwd(const float* dxn, int64_t ldd, const float* x0, int64_t ld0, const float* u,
                         int64_t ldu, float* du, int64_t lddu, float* dx0_acc, int64_t ldx0, float* dxl,
                         int64_t ldxl, int64_t rows, int32_t cols, int device, void* stream);
/* The same pass for a stack of cross layers walked from the last to the first with ONE dX0 buffer:
 * `accumulate` = 0 writes dX0 (last layer: no zero fill), != 0 adds to it; `identity` != 0 also adds
 * dXn itself (first layer, whose x_l is x0); dxl may be NULL (inner layers: the grad-X GEMM then
 * accumulates straight into the buffer dXn lives in). */
int rx_cross_combine_bwd_stack(const float* dxn, int64_t ldd, const float* x0, int64_t ld0, const float* u,
                               int64_t ldu, float* du, int64_t lddu, float* dx0_acc, int64_t ldx0,
                               float* dxl, int64_t ldxl, int64_t rows, int32_t cols, int accumulate,
                               int identity, int device, void* stream);

/* ------------------------------------------------------------------ row-sharded exchange helpers
 * No counterpart in the reference (single device).  Row r of a table lives on rank r % G at
 * local row r / G.  Bucketise a rank's index list by owner so one all-to-all (issued by the
 * host through torch.distributed / RCCL) can ship it:
 *   counts [G] int64        number of indices owned by each rank
 *   send_local [n] int64    local row ids (r / G) grouped by owner, original order kept
 *   perm [n] int32          perm[j] = position in the input of the j-th sent index
 * d_n_valid (may be NULL): device count of the leading entries of d_idx that are real (the distinct
 * rows of rx_sort_dedup, whose count never visits the host); the rest is padding that is counted for
 * no rank and sorts behind every sent index.
 * rx_unpermute_rows scatters received rows back: out[perm[j], :] = rows[j, :];
 * rx_permute_rows gathers gradient rows into send order: out[j, :] = rows[perm[j], :].
 * d_n_valid (may be NULL) on both: device count of the leading entries of perm that are real. */
size_t rx_bucketize_workspace_bytes(int64_t n, int32_t G);
int rx_bucketize_by_owner(const int64_t* d_idx, int64_t n, const int32_t* d_n_valid, int32_t G,
                          int64_t n_rows_global, int64_t* d_counts, int64_t* d_send_local,
                          int32_t* d_perm, void* ws, size_t ws_bytes, int32_t* err_flag, int device,
                          void* stream);
int rx_unpermute_rows(const float* d_rows, const int32_t* d_perm, int64_t n,
                      const int32_t* d_n_valid, int32_t dim, float* d_out, int device, void* stream);
int rx_permute_rows(const float* d_rows, const int32_t* d_perm, int64_t n, const int32_t* d_n_valid,
                    int32_t dim, float* d_out, int device, void* stream);
/* Fixed-capacity variant of the exchange (no host-side split sizes, so nothing in the step waits
 * for the host): the owner-grouped list of rx_bucketize_by_owner is packed into G slots of
 * `capacity` entries, d_packed [G * capacity] (padding slot i of a group names local row
 * i % pad_rows, pad_rows <= every owner's local height; its gradient slot stays zero, so that row
 * is left bit-identical), and d_src_of[p] = packed slot of list position p.  A group larger than `capacity` sets
 * RX_ERR_CAPACITY in *err_flag.  Rows / gradients travel in the same layout and are read /
 * written in list order with rx_permute_rows / rx_unpermute_rows (perm = d_src_of,
 * d_n_valid = the list's device-side length). */
int rx_pack_by_owner(const int64_t* d_send_local, const int32_t* d_perm, const int64_t* d_counts,
                     int32_t G, int64_t n, int64_t capacity, int64_t pad_rows, int64_t* d_packed,
                     int32_t* d_src_of, int32_t* err_flag, int device, void* stream);
/* Index glue of the sharded FM step, one launch each.  rx_fm_flat_rows: the batch's occurrence list
 * d_out [B * fields] from d_sparse [B, fields] (field-local indices) and d_offsets [fields], all
 * (sample, field < fields - 1) entries first, then the last field's B entries (the order of rx_fm2_fwd's
 * ex_idx / item_idx split; recsys/models/matrix_factorization/factorization_machine.py:33-41 adds the
 * same offsets).  rx_compose_slots: d_slot_of [n_list] = d_src_of widened to 64 bits and
 * d_occ_slot [n_occ] = d_src_of[d_inverse[i]] (occurrence -> distinct row -> exchange slot). */
int rx_fm_flat_rows(const int64_t* d_sparse, const int64_t* d_offsets, int64_t B, int32_t fields,
                    int64_t* d_out, int device, void* stream);
int rx_compose_slots(const int32_t* d_src_of, const int32_t* d_inverse, int64_t n_list, int64_t n_occ,
                     int64_t* d_slot_of, int64_t* d_occ_slot, int device, void* stream);
/* The gradient send buffer of that layout: d_rows [G * capacity, dim] (dim a multiple of 4) and
 * the companion table's d_companion [G * capacity] (may be NULL).  Zeroes the padding slots only
 * (slot offset >= d_counts[owner], the counts of rx_bucketize_by_owner); the used slots are written
 * by the reduce (rx_segreduce_fm, RX_SEG_DENSE), so the buffer needs no whole fill.  Part of the
 * model-parallel exchange the reference's single-process loop (recsys/utils.py:87-107) does not have. */
int rx_zero_unused_slots(const int64_t* d_counts, int32_t G, int64_t capacity, int32_t dim,
                         float* d_rows, float* d_companion, int device, void* stream);

/* ------------------------------------------------------------------ evaluation
 * Top-k item codes per user from list-wise scores, replacing the host numpy
 * argsort[:, ::-1][:, :k] + take_along_axis of recsys/utils.py:40-58.  Ties resolve like a
 * stable ascending argsort read backwards: among equal scores the LARGER column comes first.
 * scores [rows, n] fp32, codes [rows, n] int64 or NULL (then column ids are returned). */
int rx_topk_codes(const float* d_scores, const int64_t* d_codes, int64_t rows, int32_t n,
                  int32_t k, int64_t* d_out, int device, void* stream);

/* ------------------------------------------------------------------ dense layers, bf16 activations in HBM
 * The Linear(+ReLU) stacks and DCN-v2 cross layers of recsys_amd.models.fields (a14; the towers of
 * recsys/models/neuralcf/deep_factorization_machine.py:6-24 in their bf16 option) with every
 * activation that only feeds matrix products stored as bfloat16 (uint16_t bit patterns, round to
 * nearest even): the products round their operands to bf16 anyway, so the values are those of
 * rx_linear_*(bf16 = 1) on fp32 storage while each operand moves half the bytes.  fp32
 * accumulation, fp32 master weights; W16 / W16T are per-use bf16 copies of a weight [N, K] made
 * by rx_to_bf16 ([N, K], and transposed [K, N] for the input gradient), rows padded to a multiple
 * of 8 elements.  Every bf16 matrix has 16-byte addressable rows (leading dimension a multiple of
 * 8 elements) whose padding columns hold ZEROS: K and N need only be multiples of 4, a 16-byte
 * piece that straddles the end of a row reads zeros and writes zeros there.  N == 1 (the tower's
 * last layer, K <= 512) takes matrix-vector kernels that read the fp32 weights `W`.
 *   rx_to_bf16         dst = bf16(src) for an fp32 [rows, cols] matrix; transpose != 0 writes
 *                      dst[c, r] (weight matrices)
 *   rx_linear16_fwd    Y = act(X16 W16^T + b), Y fp32 or bf16 (out_bf16)
 *   rx_linear16_bwd_x  dX = dY16 W (through W16T); fp32 output (accumulate: dX += ...), or bf16
 *                      output with an optional gate16 [M, K] (the bf16 output of the ReLU layer
 *                      in front: entries with gate <= 0 become 0, i.e. that layer's ReLU backward
 *                      fused into this epilogue); gate16 with an fp32 output is taken by the
 *                      one-column layer only
 *   rx_linear16_bwd_w  dW = dY16^T X16 (slabs over the batch, summed in slab order), db = column
 *                      sums of dY16 (may be NULL)
 *   rx_cross16_fwd     x_next = x0 * u + x_l in fp32 plus its bf16 copy (operand of the next products)
 *   rx_cross16_bwd     rx_cross_combine_bwd_stack with du = d * x0 written as bf16 only
 *   rx_cross16_layer_fwd  one whole cross layer in one launch: u = X16 W16^T + b (kept, the backward
 *                      needs it) and, in the same epilogue, x_next = x0 * u + x_l with its bf16 copy;
 *                      bit for bit rx_linear16_fwd followed by rx_cross16_fwd */
/* BatchNorm passes of a tower that keeps its activations in bf16: rx_bn_apply writing bf16 only, and
 * rx_bn_bwd reading that bf16 output as its ReLU / dropout gate and writing its gradient as bf16. */
int rx_bn_apply16(const float* d_x, int64_t ldx, int64_t M, int32_t C, const float* d_mean,
                  const float* d_rstd, const float* d_gamma, const float* d_beta, int relu,
                  const uint8_t* d_keep, float keep_scale, uint16_t* d_out16, int64_t ldo, int device,
                  void* stream);
int rx_bn_bwd16(const float* d_dout, int64_t ldd, const float* d_x, int64_t ldx, const uint16_t* d_out16,
                int64_t ldo, int64_t M, int32_t C, const float* d_mean, const float* d_rstd,
                const float* d_gamma, int gate, float keep_scale, int batch_stats, uint16_t* d_dx16,
                int64_t lddx, float* d_dgamma, float* d_dbeta, void* ws, size_t ws_bytes, int device,
                void* stream);
int rx_to_bf16(const float* src, int64_t ld_src, uint16_t* dst, int64_t ld_dst, int64_t rows, int32_t cols,
               int transpose, int device, void* stream);
int rx_linear16_fwd(const uint16_t* X16, int64_t ldx, const uint16_t* W16, const float* W, const float* b,
                    void* Y, int64_t ldy, int64_t M, int32_t N, int32_t K, int act, int out_bf16, int device,
                    void* stream);
int rx_linear16_bwd_x(const uint16_t* dY16, int64_t ldd, const uint16_t* W16T, const float* W, void* dX, int64_t ldx,
                      const uint16_t* gate16, int64_t ldg, int64_t M, int32_t N, int32_t K, int accumulate,
                      int out_bf16, int device, void* stream);
size_t rx_linear16_bwd_w_workspace_bytes(int64_t M, int32_t N, int32_t K);
int rx_linear16_bwd_w(const uint16_t* dY16, int64_t ldd, const uint16_t* X16, int64_t ldx, float* dW, float* db,
                      int64_t M, int32_t N, int32_t K, void* ws, size_t ws_bytes, int device, void* stream);
int rx_cross16_fwd(const float* x0, int64_t ld0, const float* u, int64_t ldu, const float* xl, int64_t ldl,
                   float* xn, int64_t ldn, uint16_t* xn16, int64_t ldn16, int64_t rows, int32_t cols, int device,
                   void* stream);
int rx_cross16_layer_fwd(const uint16_t* X16, int64_t ldx, const uint16_t* W16, const float* b, const float* x0,
                         int64_t ld0, const float* xl, int64_t ldl, float* u, int64_t ldu, float* xn, int64_t ldn,
                         uint16_t* xn16, int64_t ldn16, int64_t M, int32_t d, int device, void* stream);
int rx_cross16_bwd(const float* dxn, int64_t ldd, const float* x0, int64_t ld0, const float* u, int64_t ldu,
                   uint16_t* du16, int64_t lddu, float* dx0_acc, int64_t ldx0, float* dxl, int64_t ldxl,
                   int64_t rows, int32_t cols, int accumulate, int identity, int device, void* stream);

/* ------------------------------------------------------------------ batching
 * Negative sampling of one epoch, replacing the host numpy sequence of recsys/dataset.py:165-187
 * (rows.repeat(n_neg); randint(0, pool); negative_samples[rows, draws]; hstack with the positives;
 * labels [1, 0, ..., 0] in float64): for every positive p = (users[p], items[p]) the row
 * out_items[p] = [items[p], pool[users[p], r_1], ..., pool[users[p], r_n_neg]] with r_j uniform on
 * [0, pool_size), drawn with replacement, and out_labels[p] = [1, 0, ..., 0] (may be NULL).
 * The draws are a pure function of (seed, p, j): r_j = (hi32(mix(seed, p * n_neg + j - 1)) *
 * pool_size) >> 32 with mix(s, c) the splitmix64 finaliser of s + (c + 1) * 0x9E3779B97F4A7C15 --
 * the same distribution as the reference's numpy draws, not the same stream.  pool is
 * [n_users, pool_size] int64; a user id outside [0, n_users) sets RX_ERR_INDEX and yields code 0. */
int rx_sample_negatives(const int64_t* d_users, const int64_t* d_items, const int64_t* d_pool,
                        int64_t n, int64_t n_users, int32_t pool_size, int32_t n_neg, uint64_t seed,
                        int64_t* d_out_items, double* d_out_labels, int32_t* err_flag, int device,
                        void* stream);

/* The loop of recsys/utils.py:93-106 (`for inputs in train_dl:` ... `train_loss += loss.item()`) for a
 * step that is replayed as a HIP graph: recorded at the end of the step, one launch
 *   - adds the step's loss (d_loss: one double, or one float when loss_f64 == 0; may be NULL together
 *     with d_running) to d_running[0],
 *   - copies batch *d_cursor + ahead of every epoch tensor (rows [(c + ahead) * rows_per_batch,
 *     (c + ahead + 1) * rows_per_batch) of src[t], row_bytes[t] bytes per row, a multiple of 8) into
 *     dst[t], the static batch a recorded step reads, unless that batch index is n_batches or beyond
 *     (ahead = 1: one static batch, refilled for the next step; ahead = 2: two static batches used in
 *     turn, so that the index preparation of the next batch can run beside the current step),
 *   - and advances *d_cursor by one.
 * d_ticket: one zero-initialised uint32 owned by this call sequence (the launch leaves it at zero).
 * Launches that share a cursor must not overlap. */
#define RX_BATCH_MAX_TENSORS 8
typedef struct rx_batch_plan {
  const void* src[RX_BATCH_MAX_TENSORS];
  void* dst[RX_BATCH_MAX_TENSORS];
  int64_t row_bytes[RX_BATCH_MAX_TENSORS];
  int32_t count;
  int64_t rows_per_batch;
  int64_t n_batches;
  int64_t ahead;            /* 0 is read as 1 */
} rx_batch_plan;
int rx_epoch_advance(const rx_batch_plan* plan, int64_t* d_cursor, uint32_t* d_ticket, const void* d_loss,
                     int loss_f64, double* d_running, int device, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RECSYS_HIP_H */
