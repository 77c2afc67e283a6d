// winograd43_probe.hip — feasibility probe (NOT part of the library): a fused Winograd F(4x4,3x3) kernel for the res4 3x3
// shape of the DeeperCut net at 544x736 (NB x 34x46 pixels, 256 -> 256 channels, pad 1, fp32), on the lines of the library's
// F(2x2,3x3) kernel, to see what 36 products per 16 outputs (instead of 16 per 4) buy once the 6x6 transforms,
// the 6-wave workgroup and the coarser tile grid are paid for.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DNB=8] tools/probes/winograd43_probe.hip -o /tmp/w43 && /tmp/w43
//
// Workgroup = 4 x 4 tiles (ONE 16-tile fragment of v_mfma_f32_16x16x4_f32 = 16 x 16 output pixels) x 16 output channels,
// 6 waves = the 6 rows i of the 6x6 transform; wave i owns the positions (i, 0..5): 6 accumulators.  The 18 x 18 input
// pixels of the block are staged once per 16 channels in LDS (two buffers, one barrier per stage); a lane (tile, 4 channels)
// reads the four patch rows its transform row combines, forms B^T d B in registers and issues 24 MFMAs per stage.
// Transformed filters (36/9 of the filter bytes) are streamed from global memory in fragment order, one stage ahead.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef NB
#define NB 1
#endif
constexpr int H = 34, W = 46, C = 256, K = 256;
constexpr int TY = (H + 3) / 4, TX = (W + 3) / 4;
#ifndef NCO
#define NCO 2
#endif
constexpr int BT = 4, BN = 16 * NCO;  // NCO 16-channel output slices per workgroup: every transformed value feeds NCO MFMAs
constexpr int NBY = (TY + BT - 1) / BT, NBX = (TX + BT - 1) / BT;
constexpr int RH = 4 * BT + 2, RW = 4 * BT + 2;  // 18 x 18 staged pixels
constexpr int KC = 16;                            // channels per stage
constexpr int PSTR = KC + 4;                      // floats per staged pixel: 4*PSTR = 16 (mod 64 banks)
constexpr int NTH = 384;
constexpr int NLD = (RH * RW * (KC / 4) + NTH - 1) / NTH;
constexpr int STG = RH * RW * PSTR + 16;
// skew per tile row: the 16 tiles of a fragment (4 x 4) then read 16 distinct 16-byte bank groups
__device__ __forceinline__ int rowbase(int row) { return row * RW * PSTR + 4 * (row >> 2); }  // monotonic: rows never overlap

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

// up: packed transformed filters [K/16][6 i][C/16][6 j][64 lanes][4]
__global__ __launch_bounds__(NTH) void wino43_kernel(const float* __restrict__ x, const float* __restrict__ up, float* __restrict__ y, long long* dbg) {
  __shared__ __attribute__((aligned(16))) float stage[2][STG];
  float (*part)[4][4][64] = reinterpret_cast<float (*)[4][4][64]>(&stage[0][0]);  // [i][b][r][lane]
  static_assert(sizeof(float) * 6 * 4 * 4 * 64 <= sizeof(stage), "partials fit the ring");
  const int t = threadIdx.x, lane = t & 63, i = t >> 6;
  auto stamp = [&](int slot) {
    if (dbg && lane == 0) dbg[((long)blockIdx.x * 6 + i) * 4 + slot] = (long long)__builtin_readcyclecounter();
  };
  stamp(0);
  constexpr int nblk = NB * NBY * NBX;
  const int nt = blockIdx.x / nblk, blk = blockIdx.x % nblk;
  const int n = blk / (NBY * NBX), brem = blk % (NBY * NBX);
  const int by = brem / NBX, bx = brem % NBX;
  const int oy0 = 4 * BT * by - 1, ox0 = 4 * BT * bx - 1;
  const int kg = lane >> 4, tile = lane & 15, ty = tile >> 2, tx = tile & 3;
  // transform row i as (dE + sg dP) + kp (dQ + tu dR)
  const int rE = i == 5 ? 5 : 4;
  const int rP = (i == 0 || i == 3 || i == 4) ? 2 : 3;
  const float sg = (i == 0 || i == 5) ? -5.f : (i == 1 ? 1.f : -1.f);
  const int rQ = i == 0 ? 0 : (i == 5 ? 1 : (i <= 2 ? 2 : 3));
  const int rR = i == 0 ? 0 : 1;
  const float tu = (i == 0 || i == 5) ? 0.f : (i == 1 ? 1.f : -1.f);
  const float kp = (i == 0 || i == 5) ? 4.f : (i <= 2 ? -4.f : (i == 3 ? 2.f : -2.f));
  const float* xn = x + (long)n * H * W * C;
  int gofs[NLD], sofs[NLD];
#pragma unroll
  for (int q = 0; q < NLD; ++q) {
    const int e = t + q * NTH;
    const int pix = e / (KC / 4), cq = e % (KC / 4);
    const int py = pix / RW, px = pix % RW;
    const int iy = oy0 + py, ix = ox0 + px;
    const bool ok = pix < RH * RW && iy >= 0 && ix >= 0 && iy < H && ix < W;
    gofs[q] = ok ? (iy * W + ix) * C + cq * 4 : -1;
    sofs[q] = pix < RH * RW ? rowbase(py) + px * PSTR + cq * 4 : -1;
  }
  const int cb = 4 * tx * PSTR + kg * 4;
  const int ofsE = rowbase(4 * ty + rE) + cb, ofsP = rowbase(4 * ty + rP) + cb;
  const int ofsQ = rowbase(4 * ty + rQ) + cb, ofsR = rowbase(4 * ty + rR) + cb;
  f32x4 acc[NCO][6];
#pragma unroll
  for (int h = 0; h < NCO; ++h)
#pragma unroll
    for (int j = 0; j < 6; ++j) acc[h][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* ub = up + ((long)(nt * NCO * 6 + i) * (C / 16)) * (6 * 64 * 4) + lane * 4;
  constexpr long UH = 6L * (C / 16) * (6 * 64 * 4);  // floats between two 16-channel output slices
  f32x4 g[NLD], b[NCO][6];
  auto gload = [&](int Kq) {
#pragma unroll
    for (int q = 0; q < NLD; ++q) g[q] = gofs[q] >= 0 ? *reinterpret_cast<const f32x4*>(xn + gofs[q] + Kq * KC) : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int q = 0; q < NLD; ++q)
      if (sofs[q] >= 0) *reinterpret_cast<f32x4*>(&stage[buf][sofs[q]]) = g[q];
  };
  auto bload = [&](int k16) {
#pragma unroll
    for (int h = 0; h < NCO; ++h)
#pragma unroll
      for (int j = 0; j < 6; ++j) b[h][j] = *reinterpret_cast<const f32x4*>(ub + h * UH + ((long)k16 * 6 + j) * 256);
  };
  constexpr int NS = C / KC;
  gload(0);
  sstore(0);
  if (NS > 1) gload(1);
  stamp(1);
  auto step = [&](int Kq, auto u_tag) {
    constexpr int U = decltype(u_tag)::value;
    __syncthreads();  // stage Kq complete; every wave is done with stage Kq-1 = the buffer stage Kq+1 goes into
    // all memory requests of the step go out first and are consumed one step later: stage Kq+1 (requested during step
    // Kq-1) is written to LDS, stage Kq+2 and the filters of step Kq+1 are requested
    if (Kq + 1 < NS) sstore(U ^ 1);
    if (Kq + 2 < NS) gload(Kq + 2);
    bload(Kq);  // this step's filters: their round trip runs under the LDS reads and the transform below
    __builtin_amdgcn_sched_barrier(0);
    f32x4 tc[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const f32x4 dE = *reinterpret_cast<const f32x4*>(&stage[U][ofsE + c * PSTR]);
      const f32x4 dP = *reinterpret_cast<const f32x4*>(&stage[U][ofsP + c * PSTR]);
      const f32x4 dQ = *reinterpret_cast<const f32x4*>(&stage[U][ofsQ + c * PSTR]);
      const f32x4 dR = *reinterpret_cast<const f32x4*>(&stage[U][ofsR + c * PSTR]);
      tc[c] = (dE + sg * dP) + kp * (dQ + tu * dR);
    }
    f32x4 v[6];
    v[0] = 4.f * tc[0] - 5.f * tc[2] + tc[4];
    v[5] = 4.f * tc[1] - 5.f * tc[3] + tc[5];
    {
      const f32x4 p = tc[4] - 4.f * tc[2], q = tc[3] - 4.f * tc[1];
      v[1] = p + q, v[2] = p - q;
      const f32x4 r = tc[4] - tc[2], s = tc[3] - tc[1];
      v[3] = r + 2.f * s, v[4] = r - 2.f * s;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int h = 0; h < NCO; ++h)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[h][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[j][s], b[h][j][s], acc[h][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int K0 = 0; K0 < NS; K0 += 2) {
    step(K0, std::integral_constant<int, 0>{});
    if (K0 + 1 < NS) step(K0 + 1, std::integral_constant<int, 1>{});
  }
  stamp(2);
  // inverse transform: over j in registers, over i (six waves) through LDS
#pragma unroll
  for (int h = 0; h < NCO; ++h) {
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float m0 = acc[h][0][r], m1 = acc[h][1][r], m2 = acc[h][2][r], m3 = acc[h][3][r], m4 = acc[h][4][r], m5 = acc[h][5][r];
    const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
    part[i][0][r][lane] = m0 + s12 + s34;
    part[i][1][r][lane] = d12 + 2.f * d34;
    part[i][2][r][lane] = s12 + 4.f * s34;
    part[i][3][r][lane] = d12 + 8.f * d34 + m5;
  }
  __syncthreads();
  if (i < 4) {
    const int a = i;
    const int co = nt * BN + h * 16 + (lane & 15);
#pragma unroll
    for (int bq = 0; bq < 4; ++bq)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p0 = part[0][bq][r][lane], p1 = part[1][bq][r][lane], p2 = part[2][bq][r][lane];
        const float p3 = part[3][bq][r][lane], p4 = part[4][bq][r][lane], p5 = part[5][bq][r][lane];
        float v;
        if (a == 0) v = p0 + p1 + p2 + p3 + p4;
        else if (a == 1) v = (p1 - p2) + 2.f * (p3 - p4);
        else if (a == 2) v = (p1 + p2) + 4.f * (p3 + p4);
        else v = (p1 - p2) + 8.f * (p3 - p4) + p5;
        const int tl = 4 * (lane >> 4) + r;
        const int oy = 4 * (by * BT + (tl >> 2)) + a, ox = 4 * (bx * BT + (tl & 3)) + bq;
        if (oy < H && ox < W) y[(((long)n * H + oy) * W + ox) * K + co] = v;
      }
  }
  }
  stamp(3);
}

int main() {
  std::vector<float> hx((size_t)NB * H * W * C), hw((size_t)K * C * 9), hy((size_t)NB * H * W * K);
  srand(1);
  for (auto& v : hx) v = (rand() % 2001 - 1000) / 1000.f;
  for (auto& v : hw) v = (rand() % 2001 - 1000) / 20000.f;
  const double G[6][3] = {{1. / 4, 0, 0}, {-1. / 6, -1. / 6, -1. / 6}, {-1. / 6, 1. / 6, -1. / 6}, {1. / 24, 1. / 12, 1. / 6}, {1. / 24, -1. / 12, 1. / 6}, {0, 0, 1}};
  std::vector<float> up((size_t)(K / 16) * 6 * (C / 16) * 6 * 64 * 4);
  for (int co = 0; co < K; ++co)
    for (int ci = 0; ci < C; ++ci) {
      double g[3][3], tmp[6][3], U[6][6];
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) g[a][b] = hw[((size_t)co * C + ci) * 9 + a * 3 + b];
      for (int a = 0; a < 6; ++a)
        for (int b = 0; b < 3; ++b) tmp[a][b] = G[a][0] * g[0][b] + G[a][1] * g[1][b] + G[a][2] * g[2][b];
      for (int a = 0; a < 6; ++a)
        for (int b = 0; b < 6; ++b) U[a][b] = tmp[a][0] * G[b][0] + tmp[a][1] * G[b][1] + tmp[a][2] * G[b][2];
      const int nt = co / 16, col = co % 16, k16 = ci / 16, kg = (ci % 16) / 4, s = ci % 4;
      const int lane = kg * 16 + col;
      for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j)
          up[(((((size_t)nt * 6 + i) * (C / 16) + k16) * 6 + j) * 64 + lane) * 4 + s] = (float)U[i][j];
    }
  float *dx, *du, *dy;
  CK(hipMalloc(&dx, hx.size() * 4));
  CK(hipMalloc(&du, up.size() * 4));
  CK(hipMalloc(&dy, hy.size() * 4));
  CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(du, up.data(), up.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(dy, 0, hy.size() * 4));
  const int grid = NB * NBY * NBX * (K / BN);
  long long* dd;
  CK(hipMalloc(&dd, (size_t)grid * 6 * 4 * 8));
  hipLaunchKernelGGL(wino43_kernel, dim3(grid), dim3(NTH), 0, 0, dx, du, dy, dd);
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL(wino43_kernel, dim3(grid), dim3(NTH), 0, 0, dx, du, dy, dd);
  CK(hipDeviceSynchronize());
  {
    std::vector<long long> hd((size_t)grid * 6 * 4);
    CK(hipMemcpy(hd.data(), dd, hd.size() * 8, hipMemcpyDeviceToHost));
    double p[3] = {0, 0, 0};
    std::vector<double> tot;
    for (int w = 0; w < grid * 6; ++w) {
      for (int k = 0; k < 3; ++k) p[k] += (double)(hd[w * 4 + k + 1] - hd[w * 4 + k]);
      tot.push_back((double)(hd[w * 4 + 3] - hd[w * 4]));
    }
    std::sort(tot.begin(), tot.end());
    printf("per-wave total cycles: min %.0f  median %.0f  p90 %.0f  max %.0f\n", tot[0], tot[tot.size() / 2], tot[tot.size() * 9 / 10], tot.back());
    printf("mean cycles per wave: prologue %.0f | K loop %.0f (%.0f per 16-channel stage) | inverse transform + stores %.0f\n", p[0] / (grid * 6), p[1] / (grid * 6), p[1] / (grid * 6) / (C / 16), p[2] / (grid * 6));
  }
  CK(hipMemcpy(hy.data(), dy, hy.size() * 4, hipMemcpyDeviceToHost));
  double maxerr = 0, maxref = 0;
  for (int trial = 0; trial < 600; ++trial) {
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
  printf("NB %d: grid %d workgroups of 6 waves; max |err| %.3g (max |ref| %.3g)\n", NB, grid, maxerr, maxref);
  if (getenv("W43_MAP")) {  // error map of one output channel: by position inside the 4x4 tile
    const int co = 5;
    double e44[4][4] = {{0}};
    for (int oy = 0; oy < H; ++oy)
      for (int ox = 0; ox < W; ++ox) {
        double ref = 0;
        for (int ky = 0; ky < 3; ++ky)
          for (int kx = 0; kx < 3; ++kx) {
            const int iy = oy + ky - 1, ix = ox + kx - 1;
            if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
            for (int ci = 0; ci < C; ++ci) ref += (double)hx[((size_t)iy * W + ix) * C + ci] * hw[((size_t)co * C + ci) * 9 + ky * 3 + kx];
          }
        const double e = std::fabs(ref - hy[((size_t)oy * W + ox) * K + co]);
        e44[oy % 4][ox % 4] = std::max(e44[oy % 4][ox % 4], e);
      }
    for (int a = 0; a < 4; ++a) printf("  a=%d: %.3g %.3g %.3g %.3g\n", a, e44[a][0], e44[a][1], e44[a][2], e44[a][3]);
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, 0));
    for (int it = 0; it < 50; ++it) hipLaunchKernelGGL(wino43_kernel, dim3(grid), dim3(NTH), 0, 0, dx, du, dy, (long long*)nullptr);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / 50;
    printf("%.2f us per launch = %.1f TFLOP/s of direct-convolution work (library: F(2x2,3x3) 16.1 us at batch 1, 96.7 us at batch 8)\n", us,
           2.0 * NB * H * W * K * C * 9 / us / 1e6);
  }
  return 0;
}
