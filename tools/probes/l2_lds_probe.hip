// l2_lds_probe.hip — how fast can all 256 CUs pull an L2-resident stream into LDS?  (DESIGN 4.1c priced whole-block fusion at
// "10 TB/s of L2 -> LDS"; MI355X_MICROARCH.md gives ~34.5 TB/s for the L2s.)  Every workgroup sweeps the SAME buffer (2 MB: the
// float16 filters of one res4 bottleneck block; after the first touch it is resident in each XCD's 4 MB L2) `reps` times:
//   dma  : buffer_load_dwordx4 ... lds (1 KiB per wave request, `depth` requests in flight per wave, into a ring in LDS)
//   vgpr : global_load_dwordx4 into registers (the same 1 KiB per wave request), summed, no LDS
// hipcc --offload-arch=gfx950 -O3 tools/probes/l2_lds_probe.hip -o tools/probes/bin/l2_lds_probe && tools/probes/bin/l2_lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma16(i32x4 rs, unsigned lds, unsigned voff, unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds), "v"(voff), "s"(rs), "s"(soff) : "memory", "m0");
}
template <int DEPTH>
__global__ __launch_bounds__(256) void k_dma(const char* buf, unsigned bytes, int reps, float* sink) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const unsigned long long a = (unsigned long long)buf;
  const i32x4 rs = {(int)(unsigned)a, (int)((a >> 32) & 0xffffu), 0x7fffffff, 0x00020000};
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem + wave * (DEPTH * 1024);
  const unsigned nreq = bytes / 4096;  // requests per wave and sweep: the 4 waves interleave 1 KiB pieces
  unsigned slot = 0;
  for (int r = 0; r < reps; ++r)
    for (unsigned i = 0; i < nreq; ++i) {
      dma16(rs, lds0 + slot * 1024, lane * 16, (i * 4 + wave) * 1024);
      slot = slot + 1 == DEPTH ? 0 : slot + 1;
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");  // at most DEPTH-1 older requests stay in flight
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (sink && threadIdx.x == 0 && blockIdx.x == 0) sink[0] = *reinterpret_cast<float*>(smem);
}
template <int DEPTH>
__global__ __launch_bounds__(256) void k_vgpr(const char* buf, unsigned bytes, int reps, float* sink) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  f4 acc = {0, 0, 0, 0};
  const unsigned nreq = bytes / 4096;
  for (int r = 0; r < reps; ++r)
    for (unsigned i = 0; i + DEPTH <= nreq; i += DEPTH) {
      f4 v[DEPTH];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) v[d] = *reinterpret_cast<const f4*>(buf + ((size_t)((i + d) * 4 + wave) * 1024 + lane * 16));
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) acc += v[d];
    }
  if (sink && acc.x == 12345.f) sink[0] = acc.y;
}
int main() {
  const unsigned bytes = 2u << 20;
  char* buf;
  float* sink;
  hipMalloc(&buf, bytes);
  hipMalloc(&sink, 16);
  hipMemset(buf, 1, bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int reps = 40;
  auto time = [&](auto launch, const char* name, int wgs, int depth) {
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double tb = 5.0 * wgs * (double)bytes * reps / (ms * 1e-3) / 1e12;
    std::printf("%-5s %4d workgroups x 4 waves, %2d x 1 KiB in flight per wave: %6.2f TB/s aggregate (%5.1f GB/s per CU, %.1f B/clk at 2.4 GHz)\n", name, wgs,
                depth, tb, tb * 1e3 / 256, tb * 1e12 / 256 / 2.4e9);
  };
  for (int wgs : {256, 512, 1024}) {
    time([&] { hipLaunchKernelGGL(k_dma<2>, dim3(wgs), dim3(256), 4 * 2 * 1024, 0, buf, bytes, reps, sink); }, "dma", wgs, 2);
    time([&] { hipLaunchKernelGGL(k_dma<4>, dim3(wgs), dim3(256), 4 * 4 * 1024, 0, buf, bytes, reps, sink); }, "dma", wgs, 4);
    time([&] { hipLaunchKernelGGL(k_dma<8>, dim3(wgs), dim3(256), 4 * 8 * 1024, 0, buf, bytes, reps, sink); }, "dma", wgs, 8);
    time([&] { hipLaunchKernelGGL(k_dma<16>, dim3(wgs), dim3(256), 4 * 16 * 1024, 0, buf, bytes, reps, sink); }, "dma", wgs, 16);
    time([&] { hipLaunchKernelGGL(k_vgpr<4>, dim3(wgs), dim3(256), 0, 0, buf, bytes, reps, sink); }, "vgpr", wgs, 4);
    time([&] { hipLaunchKernelGGL(k_vgpr<8>, dim3(wgs), dim3(256), 0, 0, buf, bytes, reps, sink); }, "vgpr", wgs, 8);
  }
  return 0;
}
