// wino16_loop_probe.hip — the K loop of wino_h23 (csrc/wino_f16.hip) rebuilt piece by piece, to see which piece costs what at that
// kernel's occupancy (8 waves per workgroup, one workgroup per CU): per 16-channel sub-step a wave issues 8 MFMAs (32x32x16 f16), 40
// packed adds, 12 ds_read_b128 and 4 one-KiB filter-fragment loads, with one workgroup barrier per two sub-steps.  Wall clock only
// (s_memtime's rate moves with the load on this chip: tools/probes/mfma16_probe.hip).  Round 6.
//   hipcc --offload-arch=gfx950 -O3 wino16_loop_probe.hip -o wino16_loop_probe && ./wino16_loop_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f16x2 h2(unsigned v) { return __builtin_bit_cast(f16x2, v); }
__device__ __forceinline__ unsigned u(f16x2 v) { return __builtin_bit_cast(unsigned, v); }

// LDS: the patch reads (6 ds_read_b128 per unit) feed the adds; GLD: the filter fragments (4 x 1 KiB per sub-step and wave) come from
// a 2 MB buffer (L2-resident after the first pass) one sub-step ahead, else stay constant; BAR: barrier per two sub-steps;
// VALU: the 20 packed adds per unit, else the reads' first dwords are used as they are
// STG: the staging of the next 32 channels (3 activation loads per thread and stage from a 32 MB tensor, 12 packed multiplies, 3 ds_write_b128)
template <int LDS, int GLD, int BAR, int VALU, int STG = 0>
__global__ __launch_bounds__(512, 1) void k(float* out, const u32x4* __restrict__ filt, int iters) {
  __shared__ u32x4 lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = u32x4{(unsigned)i, 1u, 2u, 3u};
  __syncthreads();
  f32x16 acc[2][2][2];
  for (int m = 0; m < 2; ++m)
    for (int j = 0; j < 2; ++j)
      for (int n = 0; n < 2; ++n)
        for (int r = 0; r < 16; ++r) acc[m][j][n][r] = 0.f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const __amdgpu_buffer_rsrc_t fr = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(filt), 0, 0x7fffffff, 0x00020000);
  const unsigned fvo = (unsigned)(wave * 4096 + lane * 16);
  u32x4 ub[2][2];
  for (int j = 0; j < 2; ++j)
    for (int n = 0; n < 2; ++n) ub[j][n] = u32x4{(unsigned)lane, (unsigned)j, (unsigned)n, 7u};
  u32x4 pa[3], pb[3];
  for (int x = 0; x < 3; ++x) pa[x] = lds[lane + 64 * x], pb[x] = lds[lane + 64 * x + 512];
  int rofs = lane;
  u32x4 g[3];
  for (int q = 0; q < 3; ++q) g[q] = u32x4{1u, 2u, 3u, (unsigned)q};
  const f16x2 quarter = {(_Float16)0.25f, (_Float16)0.25f};
  const unsigned gvo = (unsigned)((blockIdx.x * 1536 + threadIdx.x) * 16) + (2u << 20);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (STG && h == 1) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          u32x4 w;
#pragma unroll
          for (int e = 0; e < 4; ++e) w[e] = u(h2(g[q][e]) * quarter);
          lds[3072 + ((threadIdx.x + 512 * q) & 1023)] = w;
        }
      }
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        f16x2 t[3][4];
#pragma unroll
        for (int x = 0; x < 3; ++x)
#pragma unroll
          for (int e = 0; e < 4; ++e) t[x][e] = VALU ? h2(pa[x][e]) - h2(pb[x][e]) : h2(pa[x][e]);
        if (LDS) {
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int x = 0; x < 3; ++x) pa[x] = lds[rofs + 64 * x + 1024 * m + 256 * h], pb[x] = lds[rofs + 64 * x + 1024 * m + 256 * h + 2048];
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          u32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = VALU ? u(j ? t[1][e] + t[2][e] : t[0][e] - t[2][e]) : u(t[j][e]);
#pragma unroll
          for (int n = 0; n < 2; ++n) {
            acc[m][j][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ub[j][n]), __builtin_bit_cast(f16x8, v), acc[m][j][n], 0, 0, 0);
            if (GLD && m == 1)
              ub[j][n] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(fr, fvo + (unsigned)(j * 2 + n) * 1024u, (unsigned)(((it * 2 + h) & 63) * 32768), 0));
          }
        }
      }
    }
    if (STG) {
#pragma unroll
      for (int q = 0; q < 3; ++q) g[q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(fr, gvo + (unsigned)q * 8192u, (unsigned)((it & 15) * 64), 0));
    }
    if (BAR) __syncthreads();
  }
  float s = 0;
  for (int m = 0; m < 2; ++m)
    for (int j = 0; j < 2; ++j)
      for (int n = 0; n < 2; ++n)
        for (int r = 0; r < 16; ++r) s += acc[m][j][n][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)pa[0][0];
}

template <typename K>
void run(const char* name, K kern, float* d, const u32x4* f) {
  const int iters = 2000;  // one iteration = two sub-steps = 16 MFMAs per wave
  hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, d, f, 20);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, d, f, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  printf("%-58s %7.3f ms  %6.1f ns per sub-step (256 ns = the MFMAs alone at 2.0 GHz)  %5.0f TFLOP/s\n", name, best, best * 1e6 / (2.0 * iters),
         256.0 * 8 * iters * 16 * 32768.0 / (best * 1e-3) / 1e12);
}

int main() {
  float* d;
  u32x4* f;
  (void)hipMalloc(&d, 256 * 512 * 4);
  (void)hipMalloc(&f, 40 << 20);
  (void)hipMemset(f, 0x3c, 40 << 20);
  run("MFMAs only", k<0, 0, 0, 0>, d, f);
  run("+ barrier per two sub-steps", k<0, 0, 1, 0>, d, f);
  run("+ 40 packed adds per sub-step", k<0, 0, 1, 1>, d, f);
  run("+ 12 ds_read_b128 per sub-step feeding the adds", k<1, 0, 1, 1>, d, f);
  run("+ 4 filter-fragment loads per sub-step (L2)", k<1, 1, 1, 1>, d, f);
  run("+ staging of the next 32 channels (global -> x 1/4 -> LDS)", k<1, 1, 1, 1, 1>, d, f);
  run("  the same without the barrier", k<1, 1, 0, 1>, d, f);
  run("  loads + barrier, no LDS reads", k<0, 1, 1, 1>, d, f);
  run("  loads only (no reads, no adds, no barrier)", k<0, 1, 0, 0>, d, f);
  return 0;
}
