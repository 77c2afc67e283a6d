// clock_probe.hip — what shader clock does the chip sustain while something else loads it?  (diagnostic, stands alone)
//
// One wave per XCD-ish (16 workgroups of one wave) spins for ~`spin_us` and reports shader cycles (s_memtime) per tick of the constant
// 100 MHz clock (s_memrealtime): the effective shader clock over the spin.  Launched every `period_ms` for `seconds`, from a process of
// its own, beside whatever load another process runs (tools/clock_under_load.sh): the fp32-MFMA "peak" of 157.3 TFLOP/s is 256 CUs x 4
// SIMDs x 256 flop per 4 cycles... at 2.4 GHz; under a dense MFMA load the power budget holds the clock below that (MI355X_MICROARCH.md,
// "DVFS give-back"), and this says by how much for THIS load.
//   usage: clock_probe [seconds=10] [period_ms=20] [spin_us=50]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#define CHECK(e)                                                                          \
  do {                                                                                    \
    hipError_t e_ = (e);                                                                  \
    if (e_ != hipSuccess) {                                                               \
      std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_));                        \
      std::exit(1);                                                                       \
    }                                                                                     \
  } while (0)

__global__ void clock_probe_kernel(long long* out, long long spin_ticks) {
  const long long r0 = (long long)__builtin_amdgcn_s_memrealtime();
  const long long c0 = (long long)__builtin_readcyclecounter();
  long long r1 = r0;
  while (r1 - r0 < spin_ticks) {
    __builtin_amdgcn_s_sleep(8);
    r1 = (long long)__builtin_amdgcn_s_memrealtime();
  }
  const long long c1 = (long long)__builtin_readcyclecounter();
  r1 = (long long)__builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) out[2 * blockIdx.x] = c1 - c0, out[2 * blockIdx.x + 1] = r1 - r0;
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? std::atof(argv[1]) : 10.0;
  const int period_ms = argc > 2 ? std::atoi(argv[2]) : 20;
  const int spin_us = argc > 3 ? std::atoi(argv[3]) : 50;
  const int nb = 16;
  long long* d = nullptr;
  CHECK(hipMalloc(&d, sizeof(long long) * 2 * nb));
  hipStream_t st;
  CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  std::vector<long long> h(2 * nb);
  std::vector<double> ghz;
  const auto t0 = std::chrono::steady_clock::now();
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
    hipLaunchKernelGGL(clock_probe_kernel, dim3(nb), dim3(64), 0, st, d, (long long)spin_us * 100);
    CHECK(hipMemcpyAsync(h.data(), d, sizeof(long long) * 2 * nb, hipMemcpyDeviceToHost, st));
    CHECK(hipStreamSynchronize(st));
    for (int b = 0; b < nb; ++b)
      if (h[2 * b + 1] > 0) ghz.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 0.1);  // cycles per 10-ns tick -> GHz
    std::this_thread::sleep_for(std::chrono::milliseconds(period_ms));
  }
  std::sort(ghz.begin(), ghz.end());
  if (ghz.empty()) return 1;
  double sum = 0;
  for (double g : ghz) sum += g;
  std::printf("clock probe: %zu samples of %d us over %.0f s: shader clock mean %.3f GHz, median %.3f, p10 %.3f, p90 %.3f, min %.3f, max %.3f\n", ghz.size(), spin_us,
              seconds, sum / ghz.size(), ghz[ghz.size() / 2], ghz[ghz.size() / 10], ghz[ghz.size() * 9 / 10], ghz.front(), ghz.back());
  return 0;
}
