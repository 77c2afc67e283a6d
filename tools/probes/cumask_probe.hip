// Which CUs / XCDs does a CU-masked stream (hipExtStreamCreateWithCUMask) run on?  Prints, per mask hypothesis, the histogram of
// XCC_ID (hardware register) over the workgroups of a long-enough kernel, and the time of a fixed amount of work.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/cumask_probe.hip -o tools/probes/bin/cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void who(unsigned* xcc, unsigned* cu, float* sink, int spin) {
  unsigned x, h;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h));
  float a = threadIdx.x;
  for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;
  if (threadIdx.x == 0) { xcc[blockIdx.x] = x & 0xf; cu[blockIdx.x] = h; }
  if (a == 12345.f) sink[0] = a;
}
int main() {
  const int G = 2048;
  unsigned *dx, *dc; float* ds;
  CK(hipMalloc(&dx, G * 4)); CK(hipMalloc(&dc, G * 4)); CK(hipMalloc(&ds, 4));
  std::vector<unsigned> hx(G), hc(G);
  auto run = [&](const char* name, hipStream_t st) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(who, dim3(G), dim3(256), 0, st, dx, dc, ds, 2000);
    CK(hipEventRecord(e0, st));
    hipLaunchKernelGGL(who, dim3(G), dim3(256), 0, st, dx, dc, ds, 20000);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(hx.data(), dx, G * 4, hipMemcpyDeviceToHost));
    int hist[16] = {0};
    for (int i = 0; i < G; ++i) hist[hx[i] & 15]++;
    printf("%-28s %7.3f ms  xcc histogram:", name, ms);
    for (int i = 0; i < 8; ++i) printf(" %4d", hist[i]);
    printf("\n");
  };
  hipStream_t s0; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  run("no mask", s0);
  for (int x = 0; x < 8; x += 3) {  // hypothesis A: mask bit i -> XCC i % 8
    unsigned m[8] = {0};
    for (int i = 0; i < 256; ++i) if (i % 8 == x) m[i / 32] |= 1u << (i % 32);
    hipStream_t s; CK(hipExtStreamCreateWithCUMask(&s, 8, m));
    char nm[64]; snprintf(nm, sizeof nm, "interleaved: xcc %d", x); run(nm, s);
  }
  for (int x = 0; x < 8; x += 3) {  // hypothesis B: mask bits [32x, 32x+32) -> XCC x
    unsigned m[8] = {0}; m[x] = 0xffffffffu;
    hipStream_t s; CK(hipExtStreamCreateWithCUMask(&s, 8, m));
    char nm[64]; snprintf(nm, sizeof nm, "blocked: word %d", x); run(nm, s);
  }
  {  // two XCDs, interleaved hypothesis
    unsigned m[8] = {0};
    for (int i = 0; i < 256; ++i) if (i % 8 < 2) m[i / 32] |= 1u << (i % 32);
    hipStream_t s; CK(hipExtStreamCreateWithCUMask(&s, 8, m)); run("interleaved: xcc 0-1", s);
  }
  return 0;
}
