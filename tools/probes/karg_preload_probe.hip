// karg_preload_probe.hip — what a launch of the convolution kernel pays for its kernel arguments, and what handing the first 16 dwords
// to the wave in SGPRs (amdgpu-kernarg-preload-count, gfx950) would give back.  Round 6, review item 3.
// The library kernel takes ONE 648-byte by-value block; its waves stamp ~0.44 us between entry and "arguments there" on every one of the
// ~153 launches of a batch-1 forward (DC_DEBUG_TIMING).  Three forms of the same small kernel (a dependent global load per thread, the
// address needs two argument words; a far field of the block is used at the end), 200 workgroups x 256 threads like a res4 launch:
//   block     the by-value block, as the library kernel
//   touch     the block + one s_load of every 64-byte line at entry (the library's DC_KARG_TOUCH)
//   preload   the first 16 dwords as scalar arguments (preloaded into SGPRs when built with the flag), the rest of the block behind them
// Each form is timed as a hipGraph of 400 back-to-back launches (per-launch time = graph time / 400) and stamps entry -> address known.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=16 karg_preload_probe.hip -o karg_preload_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

struct Block {
  const float* x;
  float* y;
  int n, stride;
  int mid[60];
  int far[98];  // 8+8+8+240+392 = 656 bytes
};
static_assert(sizeof(Block) >= 648, "as large as the library's argument block");

struct Rest {
  int mid[50];
  int far[98];
};

__device__ __forceinline__ long long now() { return (long long)__builtin_amdgcn_s_memrealtime(); }

template <bool TOUCH>
__global__ __launch_bounds__(256) void k_block(const Block b, long long* st) {
  const long long t0 = now();
  unsigned hold0 = 0;
  if (TOUCH) {
    const auto* kp = __builtin_amdgcn_kernarg_segment_ptr();
    unsigned a0, a1, a2, a3, a4;
    asm volatile("s_load_dword %0, %5, 0x0\n\ts_load_dword %1, %5, 0x40\n\ts_load_dword %2, %5, 0x80\n\ts_load_dword %3, %5, 0xc0\n\ts_load_dword %4, %5, 0x100"
                 : "=&s"(a0), "=&s"(a1), "=&s"(a2), "=&s"(a3), "=&s"(a4) : "s"(kp));
    hold0 = a0 ^ a1 ^ a2 ^ a3 ^ a4;
  }
  const int i = blockIdx.x * 256 + threadIdx.x;
  const unsigned idx = ((unsigned)i * (unsigned)b.stride) % (unsigned)b.n;
  asm volatile("" ::"v"(idx));
  const long long t1 = now();
  const float v = b.x[idx];
  b.y[i] = v + (float)b.far[90] + (hold0 == 0x12345u ? 1.f : 0.f);
  if (threadIdx.x == 0 && st) {
    st[blockIdx.x * 2] = t1 - t0;
    st[blockIdx.x * 2 + 1] = now() - t0;
  }
}

__global__ __launch_bounds__(256) void k_pre(const float* x, float* y, int n, int stride, int m0, int m1, int m2, int m3, int m4, int m5, int m6,
                                             int m7, int m8, int m9, const Rest r, long long* st) {
  const long long t0 = now();
  const int i = blockIdx.x * 256 + threadIdx.x;
  const unsigned idx = ((unsigned)i * (unsigned)stride) % (unsigned)n;
  asm volatile("" ::"v"(idx));
  const long long t1 = now();
  const float v = x[idx];
  y[i] = v + (float)(r.far[90] + m9);
  if (threadIdx.x == 0 && st) {
    st[blockIdx.x * 2] = t1 - t0;
    st[blockIdx.x * 2 + 1] = now() - t0;
  }
}

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(_e), __LINE__); return 1; } } while (0)

template <typename F>
static int timed(const char* name, F launch, long long* st, int grid) {
  hipStream_t s;
  CK(hipStreamCreate(&s));
  const int L = 400;
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int l = 0; l < L; ++l) launch(s, l == L - 1 ? st : nullptr);
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, s));
  CK(hipStreamSynchronize(s));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  double best = 1e9, sum = 0;
  const int R = 20;
  for (int r = 0; r < R; ++r) {
    CK(hipEventRecord(e0, s));
    CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    sum += ms;
    if (ms < best) best = ms;
  }
  std::vector<long long> h(grid * 2);
  CK(hipMemcpy(h.data(), st, sizeof(long long) * grid * 2, hipMemcpyDeviceToHost));
  double a = 0, b = 0;
  for (int i = 0; i < grid; ++i) a += h[2 * i], b += h[2 * i + 1];
  printf("%-8s per launch %6.3f us (best of %d graph replays of %d launches; mean %6.3f) | wave entry -> address known %5.2f us | -> value stored %5.2f us (100 MHz clock)\n", name,
         best * 1e3 / L, R, L, sum / R * 1e3 / L, a / grid / 100.0, b / grid / 100.0);
  return 0;
}

int main() {
  const int grid = 200, n = 1 << 22;
  float *x, *y;
  long long* st;
  CK(hipMalloc(&x, sizeof(float) * n));
  CK(hipMalloc(&y, sizeof(float) * grid * 256));
  CK(hipMalloc(&st, sizeof(long long) * grid * 2));
  CK(hipMemset(x, 0, sizeof(float) * n));
  Block b{};
  b.x = x, b.y = y, b.n = n, b.stride = 97;
  Rest r{};
  for (int rep = 0; rep < 2; ++rep) {
    if (timed("block", [&](hipStream_t s, long long* p) { hipLaunchKernelGGL(k_block<false>, dim3(grid), dim3(256), 0, s, b, p); }, st, grid)) return 1;
    if (timed("touch", [&](hipStream_t s, long long* p) { hipLaunchKernelGGL(k_block<true>, dim3(grid), dim3(256), 0, s, b, p); }, st, grid)) return 1;
    if (timed("preload", [&](hipStream_t s, long long* p) { hipLaunchKernelGGL(k_pre, dim3(grid), dim3(256), 0, s, (const float*)x, y, n, 97, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, r, p); }, st, grid)) return 1;
  }
  return 0;
}
