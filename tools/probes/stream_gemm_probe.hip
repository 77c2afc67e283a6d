// stream_gemm_probe.hip — feasibility probe (NOT part of the library): 1x1 convolutions of the res4 stage as a GEMM whose
// operands go from global memory straight into MFMA registers (no LDS staging, no barriers in the K loop), the way the
// Winograd kernel reads its filters.  In the library's tiles for these layers (32x64, split-K 4) every wave owns a
// distinct B slice and shares its A slice with ONE other wave, so LDS staging buys almost no reuse there.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/stream_gemm_probe.hip -o /tmp/sprobe && /tmp/sprobe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

// Y[M][N] = relu(X[M][K] . W[N][K]^T); workgroup = 32 rows x (32*WC) columns, WK waves split K; wave = one 32x32 fragment
template <int WC, int WK, int D>
__global__ __launch_bounds__(WC* WK * 64) void sgemm_stream(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y,
                                                           int M, int N, int K) {
  __shared__ float part[WK][WC][16][64];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wc = wave % WC, wk = wave / WC;
  const int tiles_n = N / (32 * WC);
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
  const int row = tm * 32 + (lane & 31), col = tn * 32 * WC + wc * 32 + (lane & 31);
  const int kq = lane >> 5;  // which half of an 8-wide k group this lane feeds (MFMA 32x32x2: k = lane / 32)
  const int klen = K / WK, k0 = wk * klen;
  const bool rv = row < M;
  const float* xa = x + (long)(rv ? row : 0) * K + k0 + 4 * kq;
  const float* wb = w + (long)col * K + k0 + 4 * kq;
  f32x16 acc = {0};
  f32x4 a[D], b[D];
  const int steps = klen / 8;
#pragma unroll
  for (int s = 0; s < D - 1; ++s) {
    a[s] = rv ? *reinterpret_cast<const f32x4*>(xa + 8 * s) : f32x4{0.f, 0.f, 0.f, 0.f};
    b[s] = *reinterpret_cast<const f32x4*>(wb + 8 * s);
  }
  for (int s0 = 0; s0 < steps; s0 += D) {
#pragma unroll
    for (int u = 0; u < D; ++u) {
      const int s = s0 + u;
      const int sp = s + D - 1;
      if (sp < steps) {
        a[(u + D - 1) % D] = rv ? *reinterpret_cast<const f32x4*>(xa + 8 * sp) : f32x4{0.f, 0.f, 0.f, 0.f};
        b[(u + D - 1) % D] = *reinterpret_cast<const f32x4*>(wb + 8 * sp);
      }
      if (s < steps) {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][q], b[u][q], acc, 0, 0, 0);
      }
    }
  }
  // split-K reduction through LDS: wave wk == 0 sums and stores
  if (WK > 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) part[wk][wc][r][lane] = acc[r];
    __syncthreads();
    if (wk != 0) return;
#pragma unroll
    for (int q = 1; q < WK; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] += part[q][wc][r][lane];
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int orow = tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (orow < M) y[(long)orow * N + tn * 32 * WC + wc * 32 + (lane & 31)] = fmaxf(acc[r], 0.f);
  }
}

template <int WC, int WK, int D>
static void run(const char* name, int M, int N, int K, double lib_us) {
  std::vector<float> hx((size_t)M * K), hw((size_t)N * K), hy((size_t)M * N);
  srand(3);
  for (auto& v : hx) v = (rand() % 2001 - 1000) / 1000.f;
  for (auto& v : hw) v = (rand() % 2001 - 1000) / 30000.f;
  float *dx, *dw, *dy;
  CK(hipMalloc(&dx, hx.size() * 4));
  CK(hipMalloc(&dw, hw.size() * 4));
  CK(hipMalloc(&dy, hy.size() * 4));
  CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  const int grid = ((M + 31) / 32) * (N / (32 * WC));
  hipLaunchKernelGGL((sgemm_stream<WC, WK, D>), dim3(grid), dim3(WC * WK * 64), 0, 0, dx, dw, dy, M, N, K);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(hy.data(), dy, hy.size() * 4, hipMemcpyDeviceToHost));
  double maxerr = 0;
  for (int trial = 0; trial < 300; ++trial) {
    const int r = rand() % M, c = rand() % N;
    double ref = 0;
    for (int k = 0; k < K; ++k) ref += (double)hx[(size_t)r * K + k] * hw[(size_t)c * K + k];
    ref = ref > 0 ? ref : 0;
    maxerr = std::max(maxerr, std::fabs(ref - hy[(size_t)r * N + c]));
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  double best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, 0));
    for (int it = 0; it < 50; ++it) hipLaunchKernelGGL((sgemm_stream<WC, WK, D>), dim3(grid), dim3(WC * WK * 64), 0, 0, dx, dw, dy, M, N, K);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = std::min(best, ms * 1e3 / 50);
  }
  printf("%-34s grid %4d  err %.2g  %.2f us = %.1f TFLOP/s   (library gather-GEMM: %.1f us)\n", name, grid, maxerr, best,
         2.0 * M * N * K / best / 1e6, lib_us);
  CK(hipFree(dx));
  CK(hipFree(dw));
  CK(hipFree(dy));
}

int main() {
  run<2, 4, 4>("res4 2a 1024->256  32x64 wk4 d4", 1564, 256, 1024, 13.0);
  run<2, 4, 6>("res4 2a 1024->256  32x64 wk4 d6", 1564, 256, 1024, 13.0);
  run<1, 8, 4>("res4 2a 1024->256  32x32 wk8 d4", 1564, 256, 1024, 13.0);
  run<4, 2, 4>("res4 2c 256->1024  32x128 wk2 d4", 1564, 1024, 256, 13.9);
  run<2, 2, 4>("res4 2c 256->1024  32x64 wk2 d4", 1564, 1024, 256, 13.9);
  run<2, 4, 4>("res4 2c 256->1024  32x64 wk4 d4", 1564, 1024, 256, 13.9);
  return 0;
}
