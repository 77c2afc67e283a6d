// winograd32_probe.hip — feasibility probe (NOT part of the library): the library's Winograd F(2x2,3x3) algorithm with
// 32 x 32 MFMA fragments (v_mfma_f32_32x32x2_f32) instead of 16 x 16 (v_mfma_f32_16x16x4_f32), on the res4 3x3 shape
// (NB x 34x46 pixels, 256 -> 256 channels, pad 1, fp32).  A workgroup = 4 x 8 tiles (ONE 32-tile fragment) x 32 output
// channels, 4 waves = the transform rows; every LDS read, every transformed value and every filter fragment load then feeds
// twice the matrix work: the question is what that buys in CU-time (forwards in flight, batches) against the halved number
// of waves (a lone batch-1 layer has 120 workgroups of 4 waves instead of 240 of 8).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DNB=8] tools/probes/winograd32_probe.hip -o /tmp/w32 && /tmp/w32
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef NB
#define NB 1
#endif
constexpr int H = 34, W = 46, C = 256, K = 256;
constexpr int TY = (H + 1) / 2, TX = (W + 1) / 2;
constexpr int BTY = 4, BTX = 8, BN = 32;
constexpr int NBY = (TY + BTY - 1) / BTY, NBX = (TX + BTX - 1) / BTX;
constexpr int RH = 2 * BTY + 2, RW = 2 * BTX + 2;  // 10 x 18 staged pixels
constexpr int KC = 32;                             // channels per stage = 4 sub-steps of 8
constexpr int PSTR = KC + 4;
constexpr int NTH = 256;
constexpr int NLD = (RH * RW * (KC / 4) + NTH - 1) / NTH;
__device__ __forceinline__ int rowbase(int row) { return row * RW * PSTR + 4 * ((row >> 1) & 1); }

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

__device__ __forceinline__ f32x2 lo2(f32x4 v) { return __builtin_shufflevector(v, v, 0, 1); }
__device__ __forceinline__ f32x2 hi2(f32x4 v) { return __builtin_shufflevector(v, v, 2, 3); }

// up: packed transformed filters [K/32][4 i][C/8][4 j][64 lanes = (k half, 32 columns)][4]
__global__ __launch_bounds__(NTH, 2) void wino32_kernel(const float* __restrict__ x, const float* __restrict__ up, float* __restrict__ y, long long* dbg) {
  __shared__ __attribute__((aligned(16))) float stage[3][RH * RW * PSTR + 8];
  float (*part)[2][16][64] = reinterpret_cast<float (*)[2][16][64]>(&stage[0][0]);  // [i][b][r][lane]: 32 KB
  static_assert(sizeof(float) * 4 * 2 * 16 * 64 <= sizeof(stage), "partials fit the ring");
  const int t = threadIdx.x, lane = t & 63, i = t >> 6;
  auto stamp = [&](int k) { if (dbg && lane == 0) dbg[((long)blockIdx.x * 4 + i) * 4 + k] = (long long)__builtin_readcyclecounter(); };
  stamp(0);
  constexpr int nblk = NB * NBY * NBX;
  const int nt = blockIdx.x / nblk, blk = blockIdx.x % nblk;
  const int n = blk / (NBY * NBX), brem = blk % (NBY * NBX);
  const int by = brem / NBX, bx = brem % NBX;
  const int oy0 = 2 * BTY * by - 1, ox0 = 2 * BTX * bx - 1;
  const int kh = lane >> 5, tile = lane & 31, r = tile >> 3, c = tile & 7;
  const int ra = i == 0 ? 0 : (i == 2 ? 2 : 1), rb = i == 0 ? 2 : (i == 1 ? 2 : (i == 2 ? 1 : 3));
  const float sb = i == 1 ? 1.f : -1.f;
  const float* xn = x + (long)n * H * W * C;
  int gofs[NLD], sofs[NLD];
#pragma unroll
  for (int q = 0; q < NLD; ++q) {
    const int e = t + q * NTH;
    const int pix = e / (KC / 4), cq = e % (KC / 4);
    const int py = pix / RW, px = pix % RW;
    const int iy = oy0 + py, ix = ox0 + px;
    const bool ok = pix < RH * RW && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    gofs[q] = ok ? (iy * W + ix) * C + cq * 4 : -1;
    sofs[q] = pix < RH * RW ? rowbase(py) + px * PSTR + cq * 4 : -1;
  }
  const int ofs_a = rowbase(2 * r + ra) + 2 * c * PSTR + kh * 4;
  const int ofs_b = rowbase(2 * r + rb) + 2 * c * PSTR + kh * 4;
  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  const float* ub = up + ((long)(nt * 4 + i) * (C / 8)) * (4 * 64 * 4) + lane * 4;
  f32x4 g[NLD], b[3][4];
  auto gload = [&](int Kq) {
#pragma unroll
    for (int q = 0; q < NLD; ++q) g[q] = gofs[q] >= 0 ? *reinterpret_cast<const f32x4*>(xn + gofs[q] + Kq * KC) : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int q = 0; q < NLD; ++q)
      if (sofs[q] >= 0) *reinterpret_cast<f32x4*>(&stage[buf][sofs[q]]) = g[q];
  };
  auto bload = [&](int slot, int k8) {
#pragma unroll
    for (int j = 0; j < 4; ++j) b[slot][j] = *reinterpret_cast<const f32x4*>(ub + ((long)k8 * 4 + j) * 256);
  };
  f32x4 da[2][4], db[2][4];
  auto lread = [&](int slot, int buf, int h) {
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
      da[slot][c4] = *reinterpret_cast<const f32x4*>(&stage[buf][ofs_a + c4 * PSTR + h * 8]);
      db[slot][c4] = *reinterpret_cast<const f32x4*>(&stage[buf][ofs_b + c4 * PSTR + h * 8]);
    }
  };
  auto compute = [&](int slot, int bslot) {
    f32x2 tl[4], th[4];
    const f32x2 sb2 = {sb, sb};
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
      tl[c4] = lo2(da[slot][c4]) + sb2 * lo2(db[slot][c4]);
      th[c4] = hi2(da[slot][c4]) + sb2 * hi2(db[slot][c4]);
    }
    f32x2 vl[4], vh[4];
    vl[0] = tl[0] - tl[2], vh[0] = th[0] - th[2];
    vl[1] = tl[1] + tl[2], vh[1] = th[1] + th[2];
    vl[2] = tl[2] - tl[1], vh[2] = th[2] - th[1];
    vl[3] = tl[1] - tl[3], vh[3] = th[1] - th[3];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(vl[j][0], b[bslot][j][0], acc[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(vl[j][1], b[bslot][j][1], acc[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(vh[j][0], b[bslot][j][2], acc[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(vh[j][1], b[bslot][j][3], acc[j], 0, 0, 0);
  };
  constexpr int NS = C / KC, NSUB = 4 * NS;  // sub-steps of 8 channels
  gload(0);
  bload(0, 0);
  bload(1, 1);
  sstore(0);
  gload(1);
  sstore(1);
  gload(2);
  __syncthreads();
  lread(0, 0, 0);
  stamp(1);
#pragma unroll
  for (int Kq = 0; Kq < NS; ++Kq) {  // fully unrolled: every ring slot is a compile-time constant
    __syncthreads();                 // buffers <= Kq+1 are complete; buffer (Kq+2)%3 is free
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int sub = 4 * Kq + h;
      __builtin_amdgcn_sched_barrier(0);
      // LDS reads of the next sub-step (next stage's first one after the last of this stage) and the filters two ahead
      if (h < 3) lread((h + 1) & 1, Kq % 3, h + 1);
      else if (Kq + 1 < NS) lread(0, (Kq + 1) % 3, 0);
      if (sub + 2 < NSUB) bload((sub + 2) % 3, sub + 2);
      __builtin_amdgcn_sched_barrier(0);
      compute(h & 1, sub % 3);
      __builtin_amdgcn_sched_barrier(0);
      if (h == 0 && Kq + 2 < NS) sstore((Kq + 2) % 3);
      if (h == 2 && Kq + 3 < NS) gload(Kq + 3);
    }
  }
  stamp(2);
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    part[i][0][e][lane] = acc[0][e] + acc[1][e] + acc[2][e];
    part[i][1][e][lane] = acc[1][e] - acc[2][e] - acc[3][e];
  }
  __syncthreads();
  const int a = i >> 1, bq = i & 1;  // this wave finalises output pixel (a, bq) of every tile
  const int co = nt * BN + (lane & 31);
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const float p0 = part[0][bq][e][lane], p1 = part[1][bq][e][lane], p2 = part[2][bq][e][lane], p3 = part[3][bq][e][lane];
    const float v = a == 0 ? p0 + p1 + p2 : p1 - p2 - p3;
    const int q = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);  // 32x32 D layout: row = (e&3) + 8*(e>>2) + 4*(lane/32), col = lane%32
    const int ty = by * BTY + (q >> 3), tx = bx * BTX + (q & 7);
    const int oy = 2 * ty + a, ox = 2 * tx + bq;
    if (ty < TY && tx < TX && oy < H && ox < W) y[(((long)n * H + oy) * W + ox) * K + co] = v;
  }
  stamp(3);
}

int main() {
  std::vector<float> hx((size_t)NB * H * W * C), hw((size_t)K * C * 9), hy((size_t)NB * H * W * K);
  srand(1);
  for (auto& v : hx) v = (rand() % 2001 - 1000) / 1000.f;
  for (auto& v : hw) v = (rand() % 2001 - 1000) / 20000.f;
  // U = G g G^T, G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
  const double G[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
  std::vector<float> up((size_t)(K / 32) * 4 * (C / 8) * 4 * 64 * 4);
  for (int co = 0; co < K; ++co)
    for (int ci = 0; ci < C; ++ci) {
      double g[3][3], tmp[4][3], U[4][4];
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) g[a][b] = hw[((size_t)co * C + ci) * 9 + a * 3 + b];
      for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 3; ++b) tmp[a][b] = G[a][0] * g[0][b] + G[a][1] * g[1][b] + G[a][2] * g[2][b];
      for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b) U[a][b] = tmp[a][0] * G[b][0] + tmp[a][1] * G[b][1] + tmp[a][2] * G[b][2];
      const int nt = co / 32, col = co % 32, k8 = ci / 8, kh = (ci % 8) / 4, s = ci % 4;
      const int lane = kh * 32 + col;
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
          up[(((((size_t)nt * 4 + i) * (C / 8) + k8) * 4 + j) * 64 + lane) * 4 + s] = (float)U[i][j];
    }
  float *dx, *du, *dy;
  CK(hipMalloc(&dx, hx.size() * 4));
  CK(hipMalloc(&du, up.size() * 4));
  CK(hipMalloc(&dy, hy.size() * 4));
  CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(du, up.data(), up.size() * 4, hipMemcpyHostToDevice));
  const int grid = NB * NBY * NBX * (K / BN);
  long long* dd;
  CK(hipMalloc(&dd, (size_t)grid * 4 * 4 * 8));
  hipLaunchKernelGGL(wino32_kernel, dim3(grid), dim3(NTH), 0, 0, dx, du, dy, dd);
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL(wino32_kernel, dim3(grid), dim3(NTH), 0, 0, dx, du, dy, dd);
  CK(hipDeviceSynchronize());
  {
    std::vector<long long> hd((size_t)grid * 4 * 4);
    CK(hipMemcpy(hd.data(), dd, hd.size() * 8, hipMemcpyDeviceToHost));
    double p[3] = {0, 0, 0};
    std::vector<double> tot;
    for (int w = 0; w < grid * 4; ++w) {
      for (int k = 0; k < 3; ++k) p[k] += (double)(hd[w * 4 + k + 1] - hd[w * 4 + k]);
      tot.push_back((double)(hd[w * 4 + 3] - hd[w * 4]));
    }
    std::sort(tot.begin(), tot.end());
    printf("per-wave total cycles: min %.0f  median %.0f  p90 %.0f  max %.0f\n", tot[0], tot[tot.size() / 2], tot[tot.size() * 9 / 10], tot.back());
    printf("mean cycles per wave: prologue %.0f | K loop %.0f (%.0f per 16 channels) | inverse transform + stores %.0f\n", p[0] / (grid * 4), p[1] / (grid * 4), p[1] / (grid * 4) / (C / 16), p[2] / (grid * 4));
  }
  CK(hipMemcpy(hy.data(), dy, hy.size() * 4, hipMemcpyDeviceToHost));
  double maxerr = 0, maxref = 0;
  for (int trial = 0; trial < 400; ++trial) {
    const int n = rand() % NB, oy = trial < 40 ? (trial % 2 ? H - 1 : 0) : rand() % H, ox = trial < 40 ? (trial % 3 ? W - 1 : 0) : rand() % W, co = rand() % K;
    double ref = 0;
    for (int ky = 0; ky < 3; ++ky)
      for (int kx = 0; kx < 3; ++kx) {
        const int iy = oy + ky - 1, ix = ox + kx - 1;
        if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
        for (int ci = 0; ci < C; ++ci) ref += (double)hx[(((size_t)n * H + iy) * W + ix) * C + ci] * hw[((size_t)co * C + ci) * 9 + ky * 3 + kx];
      }
    maxerr = std::max(maxerr, std::fabs(ref - hy[(((size_t)n * H + oy) * W + ox) * K + co]));
    maxref = std::max(maxref, std::fabs(ref));
  }
  printf("NB %d: grid %d workgroups of 4 waves; max |err| %.3g (max |ref| %.3g)\n", NB, grid, maxerr, maxref);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, 0));
    for (int it = 0; it < 50; ++it) hipLaunchKernelGGL(wino32_kernel, dim3(grid), dim3(NTH), 0, 0, dx, du, dy, (long long*)nullptr);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / 50;
    printf("%.2f us per launch = %.1f TFLOP/s of direct-convolution work (library F(2x2,3x3) with 16x16 fragments: 16.1 us at batch 1, 96.7 us at batch 8)\n", us,
           2.0 * NB * H * W * K * C * 9 / us / 1e6);
  }
  return 0;
}
