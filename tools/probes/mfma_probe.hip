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

// non-VALU fillers between MFMAs: MODE 0 = s_nop, 1 = ds_read_b128, 2 = SALU adds, 3 = global load
template <int NACC, int MODE, int FILL>
__global__ __launch_bounds__(256) void k32f(float* out, int iters, float a0, float b0) {
  __shared__ float lds[4096];
  lds[threadIdx.x] = a0;
  lds[threadIdx.x + 256] = b0;
  __syncthreads();
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x, b = b0;
  f32x4 sink = {0, 0, 0, 0};
  int sacc = iters;
  const f32x4* lp = reinterpret_cast<const f32x4*>(lds) + (threadIdx.x & 63);
  const f32x4* gp = reinterpret_cast<const f32x4*>(out) + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u % NACC], 0, 0, 0);
#pragma unroll
      for (int f = 0; f < FILL; ++f) {
        if (MODE == 0) asm volatile("s_nop 0");
        if (MODE == 1) { f32x4 v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)(size_t)lp), "n"(0)); sink = v; }
        if (MODE == 2) asm volatile("s_add_u32 %0, %0, 3" : "+s"(sacc));
        if (MODE == 3) { f32x4 v; asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(gp)); sink = v; }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (MODE == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (MODE == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  float s = sink.x + sacc;
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

template <typename K>
void run_long(const char* name, K kern, int blocks, int threads, double flop_per_mfma, float* d, int iters, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, 10, 1.f, 2.f);
  hipDeviceSynchronize();
  for (int r = 0; r < reps; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.f, 2.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double mfma = (double)blocks * threads / 64 * iters * 16.0;
    printf("%-28s rep %d: %8.1f ms  %7.1f TF/s\n", name, r, ms, mfma * flop_per_mfma / (ms * 1e-3) / 1e12);
  }
}

int main(int argc, char** argv) {
  float* d;
  hipMalloc(&d, 4096 * 1024 * 4);
  if (argc > 1) {  // sustained-load mode: ~0.2 s launches, repeated
    const double F32 = 2.0 * 32 * 32 * 2;
    run_long("32x32x2 4acc sustained", k32<4, 0>, 256, 256, F32, d, 400000, 6);
    run_long("32x32x2 4acc 2w/SIMD sustained", k32<4, 0>, 512, 256, F32, d, 200000, 4);
    return 0;
  }
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
  run("1acc +1 s_nop", k32f<1, 0, 1>, 256, 256, F32, d);
  run("1acc +4 s_nop", k32f<1, 0, 4>, 256, 256, F32, d);
  run("2acc +4 s_nop", k32f<2, 0, 4>, 256, 256, F32, d);
  run("1acc +1 ds_read_b128", k32f<1, 1, 1>, 256, 256, F32, d);
  run("1acc +2 ds_read_b128", k32f<1, 1, 2>, 256, 256, F32, d);
  run("2acc +2 ds_read_b128", k32f<2, 1, 2>, 256, 256, F32, d);
  run("1acc +4 salu", k32f<1, 2, 4>, 256, 256, F32, d);
  run("2acc +4 salu", k32f<2, 2, 4>, 256, 256, F32, d);
  run("1acc +1 global_load x4", k32f<1, 3, 1>, 256, 256, F32, d);
  run("2acc +1 global_load x4", k32f<2, 3, 1>, 256, 256, F32, d);
  run("16x16x4 1acc", k16<1>, 256, 256, F16, d);
  run("16x16x4 2acc", k16<2>, 256, 256, F16, d);
  run("16x16x4 4acc", k16<4>, 256, 256, F16, d);
  return 0;
}
