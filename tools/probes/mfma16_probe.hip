// mfma16_probe.hip — what a wave of the float16 Winograd kernel (csrc/wino_f16.hip) can expect from gfx950: the issue cost of
// v_mfma_f32_32x32x16_f16 and of the packed float16 adds of its transform, alone and mixed, at that kernel's occupancy (8 waves per
// workgroup, one workgroup per CU = two waves per SIMD), with and without a workgroup barrier.  Round 6.
//   hipcc --offload-arch=gfx950 -O3 mfma16_probe.hip -o mfma16_probe && ./mfma16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// per iteration: 8 MFMAs (8 accumulators); FILL packed adds after each MFMA; DEP: the MFMA's column operand is the result of the
// adds in front of it; BAR: one s_barrier per iteration; MF: 0 = no MFMA at all (adds only)
template <int FILL, int DEP, int BAR, int MF>
__global__ __launch_bounds__(512, 1) void k(float* out, int iters, float a0) {
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  u32x4 a = {threadIdx.x, 2u, 3u, 4u}, b = {5u, threadIdx.x, 7u, 8u};
  f16x2 t[8];
  for (int i = 0; i < 8; ++i) t[i] = f16x2{(_Float16)(a0 + i), (_Float16)threadIdx.x};
  const long long c0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MF) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc[u], 0, 0, 0);
#pragma unroll
      for (int f = 0; f < FILL; ++f) t[(u + f) & 7] = t[(u + f) & 7] + t[(u + f + 3) & 7];
      if (DEP) b[u & 3] = __builtin_bit_cast(unsigned, t[u & 7]);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (BAR) __builtin_amdgcn_s_barrier();
  }
  const long long c1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 8; ++i) {
    s += (float)t[i][0] + (float)t[i][1];
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out + 4096 * 512)[0] = c1 - c0;
}

template <typename K>
void run(const char* name, K kern, float* d, int per_iter_mfma, int per_iter_valu) {
  const int iters = 2000;
  hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, d, 10, 1.f);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, d, iters, 1.f);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  long long cyc;
  hipMemcpy(&cyc, d + 4096 * 512, 8, hipMemcpyDeviceToHost);
  const double per_it = (double)cyc / iters;
  printf("%-46s %8.1f cycles per iteration of one wave (two waves per SIMD) = %6.1f per MFMA per SIMD", name, per_it, per_iter_mfma ? per_it / (2.0 * per_iter_mfma) : 0.0);
  if (per_iter_valu) printf(", %5.1f per add per SIMD if the adds were all", per_it / (2.0 * per_iter_valu));
  printf("  | wall %.3f ms = %.2f GHz counter rate, %.0f TFLOP/s\n", ms, (double)cyc / (ms * 1e6), 256.0 * 8 * iters * per_iter_mfma * 32768.0 / (ms * 1e-3) / 1e12);
}

int main() {
  float* d;
  hipMalloc(&d, (4096 * 512 + 16) * 4);
  run("8 MFMA", k<0, 0, 0, 1>, d, 8, 0);
  run("8 MFMA + barrier", k<0, 0, 1, 1>, d, 8, 0);
  run("8 x (MFMA + 2 pk_add)", k<2, 0, 0, 1>, d, 8, 16);
  run("8 x (MFMA + 5 pk_add)", k<5, 0, 0, 1>, d, 8, 40);
  run("8 x (MFMA + 5 pk_add), operand from the adds", k<5, 1, 0, 1>, d, 8, 40);
  run("8 x (MFMA + 5 pk_add) + barrier", k<5, 0, 1, 1>, d, 8, 40);
  run("8 x (MFMA + 8 pk_add)", k<8, 0, 0, 1>, d, 8, 64);
  run("40 pk_add, no MFMA", k<5, 0, 0, 0>, d, 0, 40);
  run("64 pk_add, no MFMA", k<8, 0, 0, 0>, d, 0, 64);
  return 0;
}
