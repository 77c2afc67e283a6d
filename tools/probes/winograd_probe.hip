// winograd_probe.hip — feasibility probe (NOT part of the library): a fused Winograd F(2x2,3x3) kernel for the
// res4 3x3 shape of the DeeperCut net at 544x736 (N=1, 34x46 pixels, 256 -> 256 channels, pad 1, fp32), to see
// whether trading 2.25x fewer MFMA flops for ~2.4x more operand traffic per MFMA cycle pays on MI355X.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/winograd_probe.hip -o /tmp/wprobe && /tmp/wprobe
//
// Design under test: no LDS, no barriers in the K loop.  A workgroup = 16 Winograd tiles x 32 output channels; wave w owns
// the row i = w of the 4x4 transform positions (i, j): it transforms its own input patches in registers (each lane: one
// tile, 4 channels), reads its own transformed-filter fragments straight from global memory (pre-packed so that a wave
// load is 1 KB contiguous) and issues v_mfma_f32_16x16x4_f32; the inverse transform reduces over j in registers and over
// i (the four waves) through LDS once at the end.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int H = 34, W = 46, C = 256, K = 256;  // input = output spatial size (stride 1, pad 1)
constexpr int TY = (H + 1) / 2, TX = (W + 1) / 2, NT = TY * TX;
constexpr int BTY = 4, BTX = 8, BN = 16;       // workgroup = 4 x 8 Winograd tiles (two 16-tile MFMA fragments) x 16 output channels
constexpr int NBY = (TY + BTY - 1) / BTY, NBX = (TX + BTX - 1) / BTX;
constexpr int RH = 2 * BTY + 2, RW = 2 * BTX + 2;  // input pixels the block reads: 10 x 18
constexpr int KC = 32;                             // channels staged per barrier interval (two 16-channel MFMA sub-steps)
constexpr int PSTR = KC + 4;                       // floats per staged pixel (+ pad: 2*PSTR = 8 mod 64 banks)
constexpr int NTH = 512;                          // 8 waves: (transform row i) x (tile fragment tf)
constexpr int NLD = (RH * RW * (KC / 4) + NTH - 1) / NTH;  // float4 global loads per thread per staged step
__device__ __forceinline__ int rowbase(int row) { return row * RW * PSTR + 4 * ((row >> 1) & 1); }  // skew: tile rows r=0/1 of a fragment hit different banks

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

// up: packed transformed filters [K/16][4 i][C/16][4 j][64 lanes][4]
__global__ __launch_bounds__(NTH) void wino_kernel(const float* __restrict__ x, const float* __restrict__ up, float* __restrict__ y, long long* dbg) {
  __shared__ __attribute__((aligned(16))) float stage[3][RH * RW * PSTR + 8];
  __shared__ float part[4][2][2][4][64];  // [i][b][tf][r][lane]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  auto stamp = [&](int k) { if (dbg && lane == 0) dbg[((long)blockIdx.x * 8 + wave) * 4 + k] = (long long)__builtin_readcyclecounter(); };
  stamp(0);
  const int blk = blockIdx.x % (NBY * NBX), nt = blockIdx.x / (NBY * NBX);
  const int by = blk / NBX, bx = blk % NBX;
  const int oy0 = 2 * BTY * by - 1, ox0 = 2 * BTX * bx - 1;  // image coordinates of staged pixel (0, 0)
  const int kg = lane >> 4;
  const int i = wave & 3, tf = wave >> 2;  // transform row and tile fragment owned by this wave
  const int ra = i == 0 ? 0 : (i == 2 ? 2 : 1), rb = i == 0 ? 2 : (i == 1 ? 2 : (i == 2 ? 1 : 3));
  const float sb = i == 1 ? 1.f : -1.f;
  // staging assignment: thread -> (pixel, channel quad) of the 10 x 18 x 16 block, NLD loads per step
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
  // LDS read offsets of this lane's tile (fragment tf: tile rows 2tf, 2tf+1 of the block; 8 tile columns), per patch row
  const int r = (lane & 15) >> 3, c = lane & 7;
  const int ofs_a = rowbase(2 * (2 * tf + r) + ra) + 2 * c * PSTR + kg * 4;
  const int ofs_b = rowbase(2 * (2 * tf + r) + rb) + 2 * c * PSTR + kg * 4;
  f32x4 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* ub = up + ((long)(nt * 4 + i) * (C / 16)) * (4 * 64 * 4) + lane * 4;
  f32x4 g[NLD], b[3][4];
  auto gload = [&](int k16) {
#pragma unroll
    for (int q = 0; q < NLD; ++q) g[q] = gofs[q] >= 0 ? *reinterpret_cast<const f32x4*>(x + gofs[q] + k16 * KC) : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int q = 0; q < NLD; ++q)
      if (sofs[q] >= 0) *reinterpret_cast<f32x4*>(&stage[buf][sofs[q]]) = g[q];
  };
  auto bload = [&](int slot, int k16) {
#pragma unroll
    for (int j = 0; j < 4; ++j) b[slot][j] = *reinterpret_cast<const f32x4*>(ub + ((long)k16 * 4 + j) * 256);
  };
  f32x4 da[2][4], db[2][4];  // patch rows of the current / next step (LDS -> registers one step ahead of the MFMAs)
  auto lread = [&](int slot, int buf, int h) {
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
      da[slot][c4] = *reinterpret_cast<const f32x4*>(&stage[buf][ofs_a + c4 * PSTR + h * 16]);
      db[slot][c4] = *reinterpret_cast<const f32x4*>(&stage[buf][ofs_b + c4 * PSTR + h * 16]);
    }
  };
  auto compute = [&](int slot, int bslot) {
    f32x2 tl[4], th[4];  // packed fp32 math where the compiler finds register pairs.  Forcing all 16 operations to v_pk_*
                         // through inline asm needs explicit wait states before the MFMAs read the results (the hazard
                         // recogniser does not see into asm: without them the answers are wrong) and is slower
                         // (17.9 vs 17.2 us): the block cannot be interleaved with the MFMAs
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
    // four independent accumulators between two MFMAs on the same one (a dependent 8-pass MFMA would need s_nops)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(vl[j][0], b[bslot][j][0], acc[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(vl[j][1], b[bslot][j][1], acc[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(vh[j][0], b[bslot][j][2], acc[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(vh[j][1], b[bslot][j][3], acc[j], 0, 0, 0);
  };
  constexpr int NS = C / KC;
  // pipeline: global -> registers (3 steps ahead) -> LDS ring of 3 (2 steps ahead) -> registers (1 sub-step ahead) -> MFMA;
  // one barrier per KC = 32 channels (two MFMA sub-steps)
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
  for (int K = 0; K < NS; ++K) {  // fully unrolled: every ring slot is a compile-time constant
    __syncthreads();              // buffers <= K+1 are complete; buffer (K+2)%3 is free
    __builtin_amdgcn_sched_barrier(0);
    lread(1, K % 3, 1);
    if (2 * K + 2 < 2 * NS) bload((2 * K + 2) % 3, 2 * K + 2);
    __builtin_amdgcn_sched_barrier(0);
    compute(0, (2 * K) % 3);
    __builtin_amdgcn_sched_barrier(0);
    if (K + 2 < NS) sstore((K + 2) % 3);
    __builtin_amdgcn_sched_barrier(0);
    if (K + 1 < NS) lread(0, (K + 1) % 3, 0);
    if (2 * K + 3 < 2 * NS) bload((2 * K + 3) % 3, 2 * K + 3);
    __builtin_amdgcn_sched_barrier(0);
    compute(1, (2 * K + 1) % 3);
    __builtin_amdgcn_sched_barrier(0);
    if (K + 3 < NS) gload(K + 3);
    __builtin_amdgcn_sched_barrier(0);
  }
  stamp(2);
  // inverse transform: over j in registers (P[b] = sum_j M[i][j] A[j][b]), over i through LDS
#pragma unroll
  for (int r4 = 0; r4 < 4; ++r4) {
    part[i][0][tf][r4][lane] = acc[0][r4] + acc[1][r4] + acc[2][r4];
    part[i][1][tf][r4][lane] = acc[1][r4] - acc[2][r4] - acc[3][r4];
  }
  __syncthreads();
  const int a = (wave >> 1) & 1, bq = wave & 1;  // this wave finalises output pixel (a, bq) of the tiles of fragment tf
#pragma unroll
  for (int r4 = 0; r4 < 4; ++r4) {
    const float p0 = part[0][bq][tf][r4][lane], p1 = part[1][bq][tf][r4][lane], p2 = part[2][bq][tf][r4][lane], p3 = part[3][bq][tf][r4][lane];
    const float v = a == 0 ? p0 + p1 + p2 : p1 - p2 - p3;
    const int q = 4 * (lane >> 4) + r4;  // D layout: row (tile in fragment) = 4*(lane/16) + r, col (channel) = lane%16
    const int ty = by * BTY + 2 * tf + (q >> 3), tx = bx * BTX + (q & 7);
    const int oy = 2 * ty + a, ox = 2 * tx + bq;
    if (ty < TY && tx < TX && oy < H && ox < W) y[((long)oy * W + ox) * K + nt * BN + (lane & 15)] = v;
  }
  stamp(3);
}

int main() {
  std::vector<float> hx((size_t)H * W * C), hw((size_t)K * C * 9), hy((size_t)H * W * K);
  srand(1);
  for (auto& v : hx) v = (rand() % 2001 - 1000) / 1000.f;
  for (auto& v : hw) v = (rand() % 2001 - 1000) / 20000.f;
  // U = G g G^T, G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
  const double G[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
  std::vector<float> up((size_t)(K / 16) * 4 * (C / 16) * 4 * 64 * 4);
  for (int co = 0; co < K; ++co)
    for (int ci = 0; ci < C; ++ci) {
      double g[3][3], tmp[4][3], U[4][4];
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) g[a][b] = hw[((size_t)co * C + ci) * 9 + a * 3 + b];
      for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 3; ++b) tmp[a][b] = G[a][0] * g[0][b] + G[a][1] * g[1][b] + G[a][2] * g[2][b];
      for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b) U[a][b] = tmp[a][0] * G[b][0] + tmp[a][1] * G[b][1] + tmp[a][2] * G[b][2];
      const int nt = co / 16, col = co % 16, k16 = ci / 16, kg = (ci % 16) / 4, s = ci % 4;
      const int lane = kg * 16 + col;
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
          up[(((((size_t)nt * 4 + i) * (C / 16) + k16) * 4 + j) * 64 + lane) * 4 + s] = (float)U[i][j];
    }
  float *dx, *du, *dy;
  CK(hipMalloc(&dx, hx.size() * 4));
  CK(hipMalloc(&du, up.size() * 4));
  CK(hipMalloc(&dy, hy.size() * 4));
  CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(du, up.data(), up.size() * 4, hipMemcpyHostToDevice));
  const int grid = NBY * NBX * (K / BN);
  long long* dd;
  CK(hipMalloc(&dd, (size_t)grid * 8 * 4 * 8));
  hipLaunchKernelGGL(wino_kernel, dim3(grid), dim3(NTH), 0, 0, dx, du, dy, dd);
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL(wino_kernel, dim3(grid), dim3(NTH), 0, 0, dx, du, dy, dd);
  CK(hipDeviceSynchronize());
  {
    std::vector<long long> hd((size_t)grid * 8 * 4);
    CK(hipMemcpy(hd.data(), dd, hd.size() * 8, hipMemcpyDeviceToHost));
    double p[3] = {0, 0, 0};
    std::vector<double> tot;
    for (int w = 0; w < grid * 8; ++w) {
      for (int k = 0; k < 3; ++k) p[k] += (double)(hd[w * 4 + k + 1] - hd[w * 4 + k]);
      tot.push_back((double)(hd[w * 4 + 3] - hd[w * 4]));
    }
    std::sort(tot.begin(), tot.end());
    printf("per-wave total cycles: min %.0f  median %.0f  p90 %.0f  max %.0f\n", tot[0], tot[tot.size() / 2], tot[tot.size() * 9 / 10], tot.back());
    printf("mean cycles per wave: prologue %.0f | K loop %.0f (%.0f per 16-channel sub-step) | inverse transform + stores %.0f\n", p[0] / (grid * 8), p[1] / (grid * 8), p[1] / (grid * 8) / (C / 16), p[2] / (grid * 8));
  }
  CK(hipMemcpy(hy.data(), dy, hy.size() * 4, hipMemcpyDeviceToHost));
  double maxerr = 0, maxref = 0;
  for (int trial = 0; trial < 400; ++trial) {
    const int oy = rand() % H, ox = rand() % W, co = rand() % K;
    double ref = 0;
    for (int ky = 0; ky < 3; ++ky)
      for (int kx = 0; kx < 3; ++kx) {
        const int iy = oy + ky - 1, ix = ox + kx - 1;
        if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
        for (int ci = 0; ci < C; ++ci) ref += (double)hx[((size_t)iy * W + ix) * C + ci] * hw[((size_t)co * C + ci) * 9 + ky * 3 + kx];
      }
    maxerr = std::max(maxerr, std::fabs(ref - hy[((size_t)oy * W + ox) * K + co]));
    maxref = std::max(maxref, std::fabs(ref));
  }
  printf("grid %d workgroups; max |err| %.3g (max |ref| %.3g)\n", grid, maxerr, maxref);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, 0));
    for (int it = 0; it < 50; ++it) hipLaunchKernelGGL(wino_kernel, dim3(grid), dim3(NTH), 0, 0, dx, du, dy, (long long*)nullptr);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / 50;
    printf("%.2f us per launch = %.1f TFLOP/s of direct-convolution work (direct kernel in the library: 23.5 us)\n", us,
           2.0 * H * W * K * C * 9 / us / 1e6);
  }
  return 0;
}
