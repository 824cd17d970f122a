// Library bf16 products for the large plain tower shapes (gfx950): hipBLASLt behind the entry points of rx_gemm.hip.
//
// The hand-written gemm_h16_kernel wins on the narrow layers (432 x 432, 216 x 432, 108 x 216 at M = 65536) and carries
// the fused epilogues (ReLU gate of the layer in front, cross-layer combine, bias-gradient column sums); on wide plain
// products the library's kernels are 1.2-1.7 x faster (profiles/gemm16_vs_library.py: 1.17 PFLOP/s against 0.69 at
// 65536 x 3344 x 3344), so those go to hipblasLtMatmul -- bias + ReLU in its epilogue where the product has them,
// beta = 1 where it accumulates.  One handle per device, made by rx_init() (never inside a stream capture), and no
// workspace of its own: forward and input-gradient products ask for algorithms that need none, the weight gradient
// lends the library the workspace its caller passed (RX_LT_WORKSPACE_MB=n lets the library have n MiB of its own,
// allocated with the handle; calls that overlap on one GPU would then share it).  Plans (descriptor, layouts, algorithm) are
// cached per shape.  Everything here returns
//   0 done, 1 error (rx_last_error), -1 not taken: the caller runs its own kernel.
// RX_NO_LIBRARY_GEMM=1 in the environment switches the whole file off (A/B timing).
#include "rx_common.h"

#include <hipblaslt/hipblaslt.h>

#include <map>
#include <mutex>
#include <tuple>

namespace {

struct LtDevice {
  hipblasLtHandle_t handle = nullptr;
  void* ws = nullptr;
  size_t ws_bytes = 0;
  bool tried = false;
};

struct LtPlan {
  hipblasLtMatmulDesc_t desc = nullptr;
  hipblasLtMatrixLayout_t a = nullptr, b = nullptr, d = nullptr;
  hipblasLtMatmulAlgo_t algo;
  size_t ws_need = 0;
  bool ok = false;
};

// (device, m, n, k, lda, ldb, ldd, transA, transB, out bf16, epilogue)
typedef std::tuple<int, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int, int, int, int> LtKey;

std::mutex lt_mutex;
LtDevice lt_dev[64];
std::map<LtKey, LtPlan> lt_plans;

const bool lt_off = [] { const char* e = getenv("RX_NO_LIBRARY_GEMM"); return e && e[0] == '1'; }();

const bool lt_verbose = [] { const char* e = getenv("RX_LT_VERBOSE"); return e && e[0] == '1'; }();
static const size_t lt_ws_want = [] { const char* e = getenv("RX_LT_WORKSPACE_MB"); return e ? (size_t)atoi(e) << 20 : (size_t)0; }();

bool lt_capturing(hipStream_t st) {
  hipStreamCaptureStatus s = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &s) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return s != hipStreamCaptureStatusNone;
}

// the device's handle and workspace; made on first use outside a capture
LtDevice* lt_device(int device, hipStream_t st) {
  if (device < 0 || device >= 64) return nullptr;
  LtDevice& d = lt_dev[device];
  if (d.handle) return &d;
  if (d.tried || lt_capturing(st)) return nullptr;
  d.tried = true;
  if (hipblasLtCreate(&d.handle) != HIPBLAS_STATUS_SUCCESS) {
    d.handle = nullptr;
    return nullptr;
  }
  if (lt_ws_want == 0) return &d;
  if (hipMalloc(&d.ws, lt_ws_want) == hipSuccess) {
    d.ws_bytes = lt_ws_want;
  } else {
    (void)hipGetLastError();
    d.ws = nullptr;
    d.ws_bytes = 0;
  }
  return &d;
}

}  // namespace

// Called by rx_init(): handle + workspace for the device, so that recordings find them in place.
int rx_lt_init(int device) {
  if (lt_off) return 0;
  std::lock_guard<std::mutex> lock(lt_mutex);
  lt_device(device, nullptr);
  return 0;
}

// D (column-major m x n, ldd; bf16 or fp32) = op(A) (m x k) op(B) (k x n) [+ D when beta1] [+ bias over the rows of D]
// [then ReLU]; A, B bf16 column-major with leading dimensions lda, ldb.  call_ws: workspace the caller lends for this
// call (the weight gradient hands over its own: products whose output tiles alone do not fill the card are split
// over k by the library, which needs room for the pieces).
int rx_lt_matmul(int device, hipStream_t st, int transA, int transB, int64_t m, int64_t n, int64_t k, const void* A, int64_t lda,
                 const void* B, int64_t ldb, void* D, int64_t ldd, int out_bf16, const float* bias, int relu, int beta1, void* call_ws,
                 size_t call_ws_bytes) {
  if (lt_off) return -1;
  std::lock_guard<std::mutex> lock(lt_mutex);
  LtDevice* dev = lt_device(device, st);
  if (!dev) return -1;
  const int epi = (bias ? 1 : 0) | (relu ? 2 : 0) | (beta1 ? 4 : 0);
  const LtKey key(device, m, n, k, lda, ldb, ldd, transA, transB, out_bf16, epi);
  auto it = lt_plans.find(key);
  if (it == lt_plans.end()) {
    LtPlan p;
    const hipDataType dt = out_bf16 ? HIP_R_16BF : HIP_R_32F;
    const hipblasOperation_t ta = transA ? HIPBLAS_OP_T : HIPBLAS_OP_N, tb = transB ? HIPBLAS_OP_T : HIPBLAS_OP_N;
    bool good = hipblasLtMatmulDescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) == HIPBLAS_STATUS_SUCCESS;
    good = good && hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(ta)) == HIPBLAS_STATUS_SUCCESS;
    good = good && hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof(tb)) == HIPBLAS_STATUS_SUCCESS;
    hipblasLtEpilogue_t e = HIPBLASLT_EPILOGUE_DEFAULT;
    if (bias && relu) e = HIPBLASLT_EPILOGUE_RELU_BIAS;
    else if (bias) e = HIPBLASLT_EPILOGUE_BIAS;
    else if (relu) e = HIPBLASLT_EPILOGUE_RELU;
    good = good && hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &e, sizeof(e)) == HIPBLAS_STATUS_SUCCESS;
    if (bias) {
      const hipDataType bt = HIP_R_32F;
      good = good && hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof(bt)) == HIPBLAS_STATUS_SUCCESS;
      good = good && hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)) == HIPBLAS_STATUS_SUCCESS;
    }
    good = good && hipblasLtMatrixLayoutCreate(&p.a, HIP_R_16BF, transA ? k : m, transA ? m : k, lda) == HIPBLAS_STATUS_SUCCESS;
    good = good && hipblasLtMatrixLayoutCreate(&p.b, HIP_R_16BF, transB ? n : k, transB ? k : n, ldb) == HIPBLAS_STATUS_SUCCESS;
    good = good && hipblasLtMatrixLayoutCreate(&p.d, dt, m, n, ldd) == HIPBLAS_STATUS_SUCCESS;
    if (good) {
      hipblasLtMatmulPreference_t pref = nullptr;
      good = hipblasLtMatmulPreferenceCreate(&pref) == HIPBLAS_STATUS_SUCCESS;
      const uint64_t max_ws = call_ws ? call_ws_bytes : dev->ws_bytes;
      good = good && hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &max_ws, sizeof(max_ws)) == HIPBLAS_STATUS_SUCCESS;
      hipblasLtMatmulHeuristicResult_t res[1];
      int found = 0;
      good = good && hipblasLtMatmulAlgoGetHeuristic(dev->handle, p.desc, p.a, p.b, p.d, p.d, pref, 1, res, &found) == HIPBLAS_STATUS_SUCCESS;
      good = good && found > 0 && res[0].state == HIPBLAS_STATUS_SUCCESS && res[0].workspaceSize <= max_ws;
      if (good) { p.algo = res[0].algo; p.ws_need = res[0].workspaceSize; }
      if (pref) hipblasLtMatmulPreferenceDestroy(pref);
    }
    p.ok = good;
    if (lt_verbose)
      fprintf(stderr, "[rx_lt] plan m=%lld n=%lld k=%lld lda=%lld ldb=%lld ldd=%lld tA=%d tB=%d out16=%d epilogue=%d: %s\n", (long long)m, (long long)n,
              (long long)k, (long long)lda, (long long)ldb, (long long)ldd, transA, transB, out_bf16, epi, good ? "library" : "own kernel");
    it = lt_plans.emplace(key, p).first;
  }
  LtPlan& p = it->second;
  if (!p.ok) return -1;
  if (bias && hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)) != HIPBLAS_STATUS_SUCCESS) return -1;
  const float alpha = 1.f, beta = beta1 ? 1.f : 0.f;
  void* ws = call_ws ? call_ws : dev->ws;
  const size_t ws_bytes = call_ws ? call_ws_bytes : dev->ws_bytes;
  if (p.ws_need > ws_bytes) return -1;
  const hipblasStatus_t rc = hipblasLtMatmul(dev->handle, p.desc, &alpha, A, p.a, B, p.b, &beta, D, p.d, D, p.d, &p.algo, ws, ws_bytes, st);
  if (rc != HIPBLAS_STATUS_SUCCESS) RX_FAIL("hipblasLtMatmul failed with status %d (m=%lld n=%lld k=%lld)", (int)rc, (long long)m, (long long)n, (long long)k);
  return 0;
}
