// What does a D = 16 row (64 bytes) plus its 4-byte bias entry cost when they live in two arrays, and
// what when they share one 128-byte line ([rows][32] floats: row in columns 0..15, bias in column 16)?
// A lane group of 4 lanes owns a row (16 bytes per lane), 8 rows per lane group with all loads issued
// before the first use -- the access shape of the row-update kernels.  Rows are distinct, uniformly
// random over the table.  Variants: read only / read-modify-write, row only / row + bias in two
// arrays / row + bias in one line.  Prints microseconds per launch and rows per second.
//   hipcc -O3 --offload-arch=gfx950 line_probe.hip -o line_probe && ./line_probe [n_rows_touched] [table_rows]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <random>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int PER = 8;

// MODE 0: row only; 1: row + bias in two arrays; 2: row + bias in one line (ld = 32); 3: as 2, but the
// write covers the whole second half of the line (bias + 15 padding floats) with 16-byte stores
template <int MODE, bool WRITE>
__global__ void __launch_bounds__(256) probe(float* __restrict__ rows, float* __restrict__ bias, const int* __restrict__ ids,
                                             int n, float* __restrict__ sink) {
  const int lane4 = threadIdx.x & 3;
  const long group = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
  const long first = group * PER;
  if (first >= n) return;
  const int ld = MODE >= 2 ? 32 : 16;
  float4 v[PER];
  float b[PER];
  long r[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) r[i] = ids[min(first + i, (long)n - 1)];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    v[i] = *reinterpret_cast<const float4*>(rows + r[i] * ld + lane4 * 4);
    b[i] = 0.f;
    if (MODE == 1 && lane4 == 0) b[i] = bias[r[i]];
    if (MODE >= 2 && lane4 == 0) b[i] = rows[r[i] * ld + 16];
  }
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    if (first + i >= n) break;
    if (WRITE) {
      float4 w = v[i];
      w.x -= 0.01f; w.y -= 0.01f; w.z -= 0.01f; w.w -= 0.01f;
      *reinterpret_cast<float4*>(rows + r[i] * ld + lane4 * 4) = w;
      if (MODE == 1 && lane4 == 0) bias[r[i]] = b[i] - 0.01f;
      if (MODE == 2 && lane4 == 0) rows[r[i] * ld + 16] = b[i] - 0.01f;
      if (MODE == 3) {
        float4 t = make_float4(lane4 == 0 ? b[i] - 0.01f : 0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(rows + r[i] * ld + 16 + lane4 * 4) = t;
      }
    } else {
      acc += v[i].x + v[i].y + v[i].z + v[i].w + b[i];
    }
  }
  if (!WRITE && acc == 123.456f) sink[0] = acc;
}

template <int MODE, bool WRITE>
static void run(const char* name, float* rows, float* bias, int* ids, int n, float* sink, hipStream_t st) {
  const int groups = (n + PER - 1) / PER;
  const int blocks = (groups * 4 + 255) / 256;
  hipEvent_t s, e;
  CK(hipEventCreate(&s)); CK(hipEventCreate(&e));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe<MODE, WRITE>), dim3(blocks), dim3(256), 0, st, rows, bias, ids + (size_t)(i % 4) * n, n, sink);
  CK(hipStreamSynchronize(st));
  const int reps = 20;
  CK(hipEventRecord(s, st));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((probe<MODE, WRITE>), dim3(blocks), dim3(256), 0, st, rows, bias, ids + (size_t)(i % 4) * n, n, sink);
  CK(hipEventRecord(e, st));
  CK(hipStreamSynchronize(st));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, s, e));
  const double us = 1e3 * ms / reps;
  printf("%-34s %8.1f us  %6.1f G rows/s\n", name, us, n / us / 1e3);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 363000;
  const long R = argc > 2 ? atol(argv[2]) : 33000000L;
  printf("rows touched per launch %d, table rows %ld\n", n, R);
  float *sep, *bias, *packed, *sink;
  int* ids;
  CK(hipMalloc(&sep, (size_t)R * 16 * 4));
  CK(hipMalloc(&bias, (size_t)R * 4));
  CK(hipMalloc(&packed, (size_t)R * 32 * 4));
  CK(hipMalloc(&sink, 256));
  CK(hipMemset(sep, 0, (size_t)R * 16 * 4));
  CK(hipMemset(bias, 0, (size_t)R * 4));
  CK(hipMemset(packed, 0, (size_t)R * 32 * 4));
  // four id lists of distinct rows: uniform, and ascending (the order the field-grouped update walks runs in is
  // neither; both ends are measured)
  std::vector<int> h((size_t)4 * n);
  std::mt19937_64 rng(1);
  for (int l = 0; l < 4; ++l) {
    std::vector<int> v(n);
    for (int i = 0; i < n; ++i) v[i] = (int)(((rng() % (R / n)) + (long)i * (R / n)));   // one row per stride: distinct
    std::shuffle(v.begin(), v.end(), rng);
    std::copy(v.begin(), v.end(), h.begin() + (size_t)l * n);
  }
  CK(hipMalloc(&ids, h.size() * 4));
  CK(hipMemcpy(ids, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  run<0, false>("read  row only", sep, bias, ids, n, sink, st);
  run<1, false>("read  row + bias, two arrays", sep, bias, ids, n, sink, st);
  run<2, false>("read  row + bias, one line", packed, bias, ids, n, sink, st);
  run<0, true>("rmw   row only", sep, bias, ids, n, sink, st);
  run<1, true>("rmw   row + bias, two arrays", sep, bias, ids, n, sink, st);
  run<2, true>("rmw   row + bias, one line", packed, bias, ids, n, sink, st);
  run<3, true>("rmw   row + bias, whole line out", packed, bias, ids, n, sink, st);
  return 0;
}
