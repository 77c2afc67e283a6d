// bw_probe.hip — practical HBM roofline for the "+shortcut" pattern: z = x + y over tensors of the res4 trunk's size
// (12512 x 1024 halves = 25.6 MB each), 16 bytes per lane, grid-stride; and a pure read / pure write pass.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/bw_probe.hip -o /tmp/bw_probe && /tmp/bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void add3(const f4* x, const f4* y, f4* z, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) z[i] = x[i] + y[i];
}
__global__ void rd(const f4* x, f4* out, long n) {
  f4 a = {0, 0, 0, 0};
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) a += x[i];
  if (a.x == 12345.f) out[0] = a;
}
__global__ void wr(f4* z, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) z[i] = f4{1, 2, 3, 4};
}
int main() {
  for (long mb : {26L, 102L, 512L, 2048L}) {
    long n = mb * 1000000L / 16;
    f4 *x, *y, *z;
    hipMalloc(&x, n * 16); hipMalloc(&y, n * 16); hipMalloc(&z, n * 16);
    hipMemset(x, 0, n * 16); hipMemset(y, 0, n * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {256, 512, 1024, 2048, 4096}) {
      float ms[3];
      for (int k = 0; k < 3; ++k) {
        for (int w = 0; w < 2; ++w) { if (k == 0) add3<<<grid, 256>>>(x, y, z, n); else if (k == 1) rd<<<grid, 256>>>(x, z, n); else wr<<<grid, 256>>>(z, n); }
        hipEventRecord(e0);
        for (int r = 0; r < 10; ++r) { if (k == 0) add3<<<grid, 256>>>(x, y, z, n); else if (k == 1) rd<<<grid, 256>>>(x, z, n); else wr<<<grid, 256>>>(z, n); }
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms[k], e0, e1);
      }
      printf("%5ld MB/tensor grid %4d: add3 %.2f TB/s (%.1f us)  read %.2f TB/s  write %.2f TB/s\n", mb, grid, 3.0 * n * 16 / (ms[0] / 10 * 1e-3) / 1e12,
             ms[0] / 10 * 1e3, 1.0 * n * 16 / (ms[1] / 10 * 1e-3) / 1e12, 1.0 * n * 16 / (ms[2] / 10 * 1e-3) / 1e12);
    }
    hipFree(x); hipFree(y); hipFree(z);
  }
  return 0;
}
