// kernels.hip — hand-written CDNA4 (gfx950) kernels of the DeeperCut forward path.
//
// conv_gemm_kernel is the hot kernel: an im2col-free gather-GEMM over channels-last (NHWC)
// activations using the exact-fp32 matrix instruction v_mfma_f32_32x32x2_f32, LDS-staged
// activation / filter tiles (double buffered, padded rows -> conflict-free ds_read_b128 operand
// fetches), and a fused epilogue (folded BatchNorm+Scale / bias, residual add, ReLU, sigmoid).
// It replaces, for this path, im2col_gpu + caffe_gpu_gemm (+ col2im_gpu for the deconvolution
// heads) and the per-layer BatchNorm/Scale/ReLU/Eltwise/Crop/Sigmoid kernels of the reference
// (src/caffe/layers/*.cu, src/caffe/util/im2col.cu, math_functions.cu) — see DESIGN.md.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace dc {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------------
// gather-GEMM convolution
// ------------------------------------------------------------------------------------------------
template <int BM, int BN, int BK, int WR, int WC, int WK>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const ConvGemmParams p) {
  static_assert(WR * WC * WK == 4, "4 waves per workgroup");
  constexpr int LDK = BK + 4;            // padded LDS row, floats (16-B aligned, bank-spread)
  constexpr int TM = BM / WR, TN = BN / WC;
  constexpr int FM = TM / 32, FN = TN / 32;
  static_assert(FM >= 1 && FN >= 1 && TM % 32 == 0 && TN % 32 == 0, "wave tile = multiples of 32x32");
  constexpr int C4 = BK / 4;             // float4 per tile row
  constexpr int RPP = 256 / C4;          // tile rows covered by one pass of the 256 threads
  constexpr int NA = BM / RPP, NBV = BN / RPP;
  static_assert(NA >= 1 && NBV >= 1, "tile too small for the loader");
  constexpr int KCH = BK / 8;            // 8-deep k chunks per tile
  static_assert(KCH % WK == 0, "k chunks must split evenly over WK");
  constexpr int TILE = (BM + BN) * LDK;  // floats per LDS stage
  static_assert((WK - 1) * BM * BN <= 2 * TILE, "split-K partials must fit in the tile buffers");

  __shared__ __attribute__((aligned(16))) float smem[2 * TILE + 2 * BM];
  long* rowoff = reinterpret_cast<long*>(smem + 2 * TILE);

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = t >> 6;
  const int wk = wave % WK;
  const int wc = (wave / WK) % WC;
  const int wr = wave / (WK * WC);

  const int tiles_n = (p.Cout + BN - 1) / BN;
  const int tile_n = blockIdx.x % tiles_n;
  const int tile_m = blockIdx.x / tiles_n;
  const int m0 = tile_m * BM;
  const int n0 = tile_n * BN;
  const int ohw = p.OH * p.OW;

  // output offsets of this tile's pixel rows (epilogue), -1 = outside the problem
  if (t < BM) {
    int m = m0 + t;
    long off = -1;
    if (m < p.M) {
      int n = m / ohw;
      int rem = m - n * ohw;
      int oy = rem / p.OW;
      int ox = rem - oy * p.OW;
      off = (long)n * p.y_img_stride + (long)oy * p.y_row_stride + (long)ox * p.y_pix_stride;
    }
    rowoff[t] = off;
  }

  // loader state: this thread stages rows (t / C4) + RPP*i, float4 column (t % C4)
  const int lrow = t / C4;
  const int lc4 = (t % C4) * 4;
  const float* arow[NA];
  int aiy[NA], axe[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    int m = m0 + lrow + RPP * i;
    if (m < p.M) {
      int n = m / ohw;
      int rem = m - n * ohw;
      int oy = rem / p.OW;
      int ox = rem - oy * p.OW;
      aiy[i] = oy * p.sy;
      axe[i] = ox * p.sx + lc4;
      arow[i] = p.x + (long)n * p.x_img_stride + (long)(oy * p.sy) * p.x_row_stride + (ox * p.sx + lc4);
    } else {
      aiy[i] = -(1 << 28);  // every tap fails the row test
      axe[i] = 0;
      arow[i] = p.x;
    }
  }
  const float* brow[NBV];
  bool bok[NBV];
#pragma unroll
  for (int j = 0; j < NBV; ++j) {
    int n = n0 + lrow + RPP * j;
    bok[j] = n < p.Cout;
    brow[j] = p.w + (long)(bok[j] ? n : 0) * p.Ktot + lc4;
  }

  f32x16 acc[FM][FN];
#pragma unroll
  for (int a = 0; a < FM; ++a)
#pragma unroll
    for (int b = 0; b < FN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int T = p.Ktot / BK;
  int tap = 0, c0 = 0, kg = 0;
  f32x4 ra[NA], rb[NBV];

  auto gload = [&]() {
    const ConvTap tp = p.taps[tap];
    const int doff = tp.dy * p.x_row_stride + tp.xoff + c0;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      int iy = aiy[i] + tp.dy;
      int xe = axe[i] + tp.xoff + c0;
      bool ok = (unsigned)iy < (unsigned)p.x_rows && (unsigned)xe < (unsigned)p.x_rowlen;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ok) v = *reinterpret_cast<const f32x4*>(arow[i] + doff);
      ra[i] = v;
    }
#pragma unroll
    for (int j = 0; j < NBV; ++j) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (bok[j]) v = *reinterpret_cast<const f32x4*>(brow[j] + kg);
      rb[j] = v;
    }
    c0 += BK;
    kg += BK;
    if (c0 >= tp.klen) {
      c0 = 0;
      ++tap;
    }
  };
  auto lstore = [&](int buf) {
    float* As = smem + buf * TILE;
    float* Bs = As + BM * LDK;
#pragma unroll
    for (int i = 0; i < NA; ++i)
      *reinterpret_cast<f32x4*>(As + (lrow + RPP * i) * LDK + lc4) = ra[i];
#pragma unroll
    for (int j = 0; j < NBV; ++j)
      *reinterpret_cast<f32x4*>(Bs + (lrow + RPP * j) * LDK + lc4) = rb[j];
  };
  auto compute = [&](int buf) {
    const float* As = smem + buf * TILE + (wr * TM + (lane & 31)) * LDK + (lane >> 5) * 4;
    const float* Bs = smem + buf * TILE + BM * LDK + (wc * TN + (lane & 31)) * LDK + (lane >> 5) * 4;
#pragma unroll
    for (int kc = wk; kc < KCH; kc += WK) {
      f32x4 av[FM], bv[FN];
#pragma unroll
      for (int a = 0; a < FM; ++a) av[a] = *reinterpret_cast<const f32x4*>(As + a * 32 * LDK + kc * 8);
#pragma unroll
      for (int b = 0; b < FN; ++b) bv[b] = *reinterpret_cast<const f32x4*>(Bs + b * 32 * LDK + kc * 8);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int a = 0; a < FM; ++a)
#pragma unroll
          for (int b = 0; b < FN; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a][s], bv[b][s], acc[a][b], 0, 0, 0);
    }
  };

  gload();
  lstore(0);
  __syncthreads();
  for (int it = 0; it < T; ++it) {
    const int buf = it & 1;
    const bool more = it + 1 < T;
    if (more) gload();
    compute(buf);
    if (more) lstore(buf ^ 1);
    __syncthreads();
  }

  // in-workgroup split-K: waves wk>0 hand their partial tiles to wave wk==0 through LDS
  if (WK > 1) {
    float* part = smem;  // tile buffers are free after the last barrier
    if (wk > 0) {
      float* dst = part + (((wk - 1) * WR * WC + wr * WC + wc) * FM * FN) * 16 * 64 + lane;
#pragma unroll
      for (int a = 0; a < FM; ++a)
#pragma unroll
        for (int b = 0; b < FN; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) dst[((a * FN + b) * 16 + r) * 64] = acc[a][b][r];
    }
    __syncthreads();
    if (wk == 0) {
#pragma unroll
      for (int q = 1; q < WK; ++q) {
        const float* src = part + (((q - 1) * WR * WC + wr * WC + wc) * FM * FN) * 16 * 64 + lane;
#pragma unroll
        for (int a = 0; a < FM; ++a)
#pragma unroll
          for (int b = 0; b < FN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] += src[((a * FN + b) * 16 + r) * 64];
      }
    }
  }
  if (wk != 0) return;

  // fused epilogue.  MFMA 32x32 C layout: col = lane&31 (channel), row = (r&3)+8*(r>>2)+4*(lane>>5).
#pragma unroll
  for (int b = 0; b < FN; ++b) {
    const int co = n0 + wc * TN + b * 32 + (lane & 31);
    const bool cok = co < p.Cout;
    const float sc = (cok && p.scale) ? p.scale[co] : 1.f;
    const float sh = (cok && p.shift) ? p.shift[co] : 0.f;
    const bool sig = co < p.sigmoid_ch;
#pragma unroll
    for (int a = 0; a < FM; ++a) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wr * TM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const long off = rowoff[row];
        if (off >= 0 && cok) {
          float v = acc[a][b][r] * sc + sh;
          if (p.resid) v += p.resid[off + co];
          if (p.relu) v = fmaxf(v, 0.f);
          if (sig) v = 1.f / (1.f + expf(-v));
          p.y[off + co] = v;
        }
      }
    }
  }
}

namespace {
struct VariantEntry {
  ConvVariant v;
  void (*kernel)(const ConvGemmParams);
  int BK;
};
#define DC_VARIANT(BM, BN, BK, WR, WC, WK)                                   \
  {                                                                          \
    {#BM "x" #BN "x" #BK "_w" #WR #WC #WK, BM, BN, WR, WC, WK},              \
        conv_gemm_kernel<BM, BN, BK, WR, WC, WK>, BK                         \
  }
const VariantEntry kVariants[] = {
    DC_VARIANT(128, 128, 32, 2, 2, 1),  // 0: big-M layers (res2/res3)
    DC_VARIANT(128, 64, 32, 2, 2, 1),   // 1
    DC_VARIANT(64, 128, 32, 2, 2, 1),   // 2
    DC_VARIANT(64, 64, 32, 2, 2, 1),    // 3
    DC_VARIANT(64, 64, 64, 2, 2, 1),    // 4
    DC_VARIANT(32, 64, 64, 1, 2, 2),    // 5: in-workgroup split-K 2
    DC_VARIANT(64, 32, 64, 2, 1, 2),    // 6
    DC_VARIANT(32, 32, 128, 1, 1, 4),   // 7: split-K 4 (tiny M*N, long K: res4/res5)
    DC_VARIANT(32, 32, 32, 1, 1, 4),    // 8: same for K segments that are only multiples of 32
};
constexpr int kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);
}  // namespace

int conv_num_variants() { return kNumVariants; }
const ConvVariant& conv_variant(int i) { return kVariants[i].v; }
int conv_variant_bk(int i) { return kVariants[i].BK; }

long conv_grid(const ConvGemmParams& p, int variant) {
  const ConvVariant& v = kVariants[variant].v;
  long tm = (p.M + v.BM - 1) / v.BM, tn = (p.Cout + v.BN - 1) / v.BN;
  return tm * tn;
}

int launch_conv_gemm(const ConvGemmParams& p, int variant, void* stream) {
  if (variant < 0 || variant >= kNumVariants) return (int)hipErrorInvalidValue;
  const VariantEntry& e = kVariants[variant];
  for (int i = 0; i < p.ntaps; ++i)
    if (p.taps[i].klen % e.BK != 0) return (int)hipErrorInvalidValue;
  long grid = conv_grid(p, variant);
  if (grid <= 0) return 0;
  hipLaunchKernelGGL(e.kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// MAX pooling (NHWC, float4 over channels)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                      int NB, int H, int W, int C, int OH, int OW, int k,
                                                      int s, int pad) {
  const int c4n = C / 4;
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)NB * OH * OW * c4n;
  if (idx >= total) return;
  int c4 = (int)(idx % c4n);
  long pix = idx / c4n;
  int ox = (int)(pix % OW);
  long t2 = pix / OW;
  int oy = (int)(t2 % OH);
  int n = (int)(t2 / OH);
  int hs = oy * s - pad, ws = ox * s - pad;
  int he = min(hs + k, H), we = min(ws + k, W);  // pooling_layer.cpp:150-155 (pad == 0 on this path)
  hs = max(hs, 0);
  ws = max(ws, 0);
  f32x4 m = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
  for (int iy = hs; iy < he; ++iy)
    for (int ix = ws; ix < we; ++ix) {
      f32x4 v = *reinterpret_cast<const f32x4*>(x + (((long)n * H + iy) * W + ix) * C + c4 * 4);
      m.x = v.x > m.x ? v.x : m.x;
      m.y = v.y > m.y ? v.y : m.y;
      m.z = v.z > m.z ? v.z : m.z;
      m.w = v.w > m.w ? v.w : m.w;
    }
  *reinterpret_cast<f32x4*>(y + (((long)n * OH + oy) * OW + ox) * C + c4 * 4) = m;
}

__global__ __launch_bounds__(256) void maxpool_scalar_kernel(const float* __restrict__ x,
                                                             float* __restrict__ y, int NB, int H, int W,
                                                             int C, int OH, int OW, int k, int s, int pad) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)NB * OH * OW * C;
  if (idx >= total) return;
  int c = (int)(idx % C);
  long pix = idx / C;
  int ox = (int)(pix % OW);
  long t2 = pix / OW;
  int oy = (int)(t2 % OH);
  int n = (int)(t2 / OH);
  int hs = oy * s - pad, ws = ox * s - pad;
  int he = min(hs + k, H), we = min(ws + k, W);
  hs = max(hs, 0);
  ws = max(ws, 0);
  float m = -3.402823466e+38f;
  for (int iy = hs; iy < he; ++iy)
    for (int ix = ws; ix < we; ++ix) {
      float v = x[(((long)n * H + iy) * W + ix) * C + c];
      m = v > m ? v : m;
    }
  y[idx] = m;
}

int launch_maxpool(const float* x, float* y, int NB, int H, int W, int C, int OH, int OW, int k, int s,
                   int pad, void* stream) {
  if (C % 4 == 0) {
    long total = (long)NB * OH * OW * (C / 4);
    if (total <= 0) return 0;
    hipLaunchKernelGGL(maxpool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, x, y, NB, H, W, C, OH, OW, k, s, pad);
  } else {
    long total = (long)NB * OH * OW * C;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(maxpool_scalar_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, x, y, NB, H, W, C, OH, OW, k, s, pad);
  }
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// stand-alone elementwise: y = act(x*a[c] + b[c] + z)
// ------------------------------------------------------------------------------------------------
__device__ inline float dc_act(float v, int relu, int sigmoid) {
  if (relu) v = fmaxf(v, 0.f);
  if (sigmoid) v = 1.f / (1.f + expf(-v));
  return v;
}

__global__ __launch_bounds__(256) void eltwise_vec4_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ z,
                                                           const float* __restrict__ a,
                                                           const float* __restrict__ b, float* __restrict__ y,
                                                           long total4, int C, int relu, int sigmoid) {
  long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += stride) {
    f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    int c = (int)((i * 4) % C);
    if (a) {
      f32x4 av = *reinterpret_cast<const f32x4*>(a + c);
      v *= av;
    }
    if (b) {
      f32x4 bv = *reinterpret_cast<const f32x4*>(b + c);
      v += bv;
    }
    if (z) v += reinterpret_cast<const f32x4*>(z)[i];
    v.x = dc_act(v.x, relu, sigmoid);
    v.y = dc_act(v.y, relu, sigmoid);
    v.z = dc_act(v.z, relu, sigmoid);
    v.w = dc_act(v.w, relu, sigmoid);
    reinterpret_cast<f32x4*>(y)[i] = v;
  }
}

__global__ __launch_bounds__(256) void eltwise_scalar_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ z,
                                                             const float* __restrict__ a,
                                                             const float* __restrict__ b,
                                                             float* __restrict__ y, long total, int C,
                                                             int relu, int sigmoid) {
  long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    float v = x[i];
    int c = (int)(i % C);
    if (a) v *= a[c];
    if (b) v += b[c];
    if (z) v += z[i];
    y[i] = dc_act(v, relu, sigmoid);
  }
}

int launch_eltwise(const float* x, const float* z, const float* a, const float* b, float* y, long total,
                   int C, int relu, int sigmoid, void* stream) {
  if (total <= 0) return 0;
  if (C % 4 == 0 && total % 4 == 0) {
    long t4 = total / 4;
    long blocks = (t4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(eltwise_vec4_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, z,
                       a, b, y, t4, C, relu, sigmoid);
  } else {
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(eltwise_scalar_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x,
                       z, a, b, y, total, C, relu, sigmoid);
  }
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// crop (NHWC)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void crop_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                   int NB, int H, int W, int C, int oh, int ow, int OH,
                                                   int OW) {
  long total = (long)NB * OH * OW * C;
  long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    int c = (int)(i % C);
    long pix = i / C;
    int ox = (int)(pix % OW);
    long t2 = pix / OW;
    int oy = (int)(t2 % OH);
    int n = (int)(t2 / OH);
    y[i] = x[(((long)n * H + (oy + oh)) * W + (ox + ow)) * C + c];
  }
}

int launch_crop(const float* x, float* y, int NB, int H, int W, int C, int oh, int ow, int OH, int OW,
                void* stream) {
  long total = (long)NB * OH * OW * C;
  if (total <= 0) return 0;
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(crop_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, y, NB, H, W,
                     C, oh, ow, OH, OW);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// NCHW <-> NHWC through a 32x32 LDS tile (both sides coalesced)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src,
                                                           float* __restrict__ dst, int C, int HW, int CP) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z;
  const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int c = c0 + ty + 8 * k, pix = p0 + tx;
    tile[ty + 8 * k][tx] = (c < C && pix < HW) ? src[((long)n * C + c) * HW + pix] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int pix = p0 + ty + 8 * k, c = c0 + tx;
    if (pix < HW && c < CP) dst[((long)n * HW + pix) * CP + c] = tile[tx][ty + 8 * k];
  }
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ src,
                                                           float* __restrict__ dst, int C, int HW, int CP,
                                                           int cbase) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z;
  const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int pix = p0 + ty + 8 * k, c = c0 + tx;
    tile[ty + 8 * k][tx] = (pix < HW && c < C) ? src[((long)n * HW + pix) * CP + cbase + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int c = c0 + ty + 8 * k, pix = p0 + tx;
    if (c < C && pix < HW) dst[((long)n * C + c) * HW + pix] = tile[tx][ty + 8 * k];
  }
}

int launch_nchw_to_nhwc(const float* src, float* dst, int NB, int C, int H, int W, int CP, void* stream) {
  int HW = H * W;
  if (NB <= 0 || HW <= 0) return 0;
  dim3 grid((HW + 31) / 32, (CP + 31) / 32, NB);
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, dst, C, HW, CP);
  return (int)hipGetLastError();
}

int launch_nhwc_to_nchw(const float* src, float* dst, int NB, int C, int H, int W, int CP, int c0,
                        void* stream) {
  int HW = H * W;
  if (NB <= 0 || HW <= 0) return 0;
  dim3 grid((HW + 31) / 32, (C + 31) / 32, NB);
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, dst, C, HW, CP, c0);
  return (int)hipGetLastError();
}

}  // namespace dc
