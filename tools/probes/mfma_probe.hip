// mfma_probe.hip — how fast does v_mfma_f32_32x32x2_f32 / 16x16x4 issue from one wave per SIMD, as a
// function of the number of independent accumulators and of interleaved non-MFMA work?
//   hipcc --offload-arch=gfx950 -O3 mfma_probe.hip -o mfma_probe && ./mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int FILL>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x, b = b0;
  float junk = a;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u % NACC], 0, 0, 0);
#pragma unroll
      for (int f = 0; f < FILL; ++f) junk = junk * 1.0001f + 0.5f;
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = junk;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a0, float b0) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) acc[u % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u % NACC], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 4; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
void run(const char* name, K kern, int blocks, int threads, double flop_per_mfma, float* d) {
  int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, 10, 1.f, 2.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.f, 2.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double waves = (double)blocks * threads / 64;
  double mfma = waves * iters * 16.0;
  double tf = mfma * flop_per_mfma / (ms * 1e-3) / 1e12;
  double ns_per_mfma_per_wave = ms * 1e6 / (iters * 16.0);
  printf("%-34s blocks=%4d thr=%d  %8.3f ms  %7.1f TF/s  %6.1f ns per MFMA per wave\n", name, blocks, threads, ms, tf, ns_per_mfma_per_wave);
}

int main() {
  float* d;
  hipMalloc(&d, 4096 * 1024 * 4);
  const double F32 = 2.0 * 32 * 32 * 2, F16 = 2.0 * 16 * 16 * 4;
  run("32x32x2 1acc 1wave/SIMD", k32<1, 0>, 256, 256, F32, d);
  run("32x32x2 2acc 1wave/SIMD", k32<2, 0>, 256, 256, F32, d);
  run("32x32x2 4acc 1wave/SIMD", k32<4, 0>, 256, 256, F32, d);
  run("32x32x2 1acc +2valu", k32<1, 2>, 256, 256, F32, d);
  run("32x32x2 1acc +8valu", k32<1, 8>, 256, 256, F32, d);
  run("32x32x2 2acc +8valu", k32<2, 8>, 256, 256, F32, d);
  run("32x32x2 4acc +8valu", k32<4, 8>, 256, 256, F32, d);
  run("32x32x2 1acc 2waves/SIMD", k32<1, 0>, 512, 256, F32, d);
  run("32x32x2 1acc 2w/SIMD +8valu", k32<1, 8>, 512, 256, F32, d);
  run("32x32x2 1acc 196 blocks", k32<1, 0>, 196, 256, F32, d);
  run("16x16x4 1acc", k16<1>, 256, 256, F16, d);
  run("16x16x4 2acc", k16<2>, 256, 256, F16, d);
  run("16x16x4 4acc", k16<4>, 256, 256, F16, d);
  return 0;
}
