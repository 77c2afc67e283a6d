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

#include <algorithm>

#include "kernels.h"

namespace dc {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------------
// gather-GEMM convolution
// ------------------------------------------------------------------------------------------------
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// Buffer addressing (V# descriptors): address = base + voffset(VGPR) + soffset(SGPR); an access whose
// voffset is >= num_records returns 0 / is dropped.  This keeps the K loop almost free of VALU work —
// which matters because on gfx950 the fp32 MFMA shares the SIMD's fp32 datapath: every VALU
// instruction issued between MFMAs is paid IN ADDITION to them (tools/probes/mfma_probe.hip:
// 143 TF/s bare, 91 TF/s with 8 VALU per MFMA, one or two waves per SIMD alike).
//   * per-thread voffsets are loop invariant, the per-tile displacement (tap, channel block) is uniform
//     and travels in soffset (SALU);
//   * zero padding = out-of-range voffset (one v_cndmask per load from a precomputed tap-validity mask).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t dc_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 dc_bload4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
constexpr unsigned kOOB = 0x80000000u;  // > any tensor size: hardware returns 0

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
  constexpr int NCH = KCH / WK;          // chunks this wave owns per tile
  static_assert(NCH >= 2 && NCH % 2 == 0, "the software pipeline needs an even number (>=2) of chunks per wave");
  constexpr int TILE = (BM + BN) * LDK;  // floats per LDS stage
  static_assert((WK - 1) * BM * BN <= 2 * TILE, "split-K partials must fit in the tile buffers");

  __shared__ __attribute__((aligned(16))) float smem[2 * TILE + 4 * BM];
  i32x4* rowinfo = reinterpret_cast<i32x4*>(smem + 2 * TILE);  // per tile row: {x byte offset, iy0, xe0, y byte offset | -1}

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

  // one pixel decode per tile row (integer divisions are VALU-expensive), shared through LDS
  if (t < BM) {
    const int m = m0 + t;
    i32x4 ri = {(int)kOOB, -(1 << 28), 0, -1};
    if (m < p.M) {
      const int ohw = p.OH * p.OW;
      const int n = m / ohw;
      const int rem = m - n * ohw;
      const int oy = rem / p.OW;
      const int ox = rem - oy * p.OW;
      ri.x = (int)(((long)n * p.x_img_stride + (long)(oy * p.sy) * p.x_row_stride + ox * p.sx) * 4);
      ri.y = oy * p.sy;
      ri.z = ox * p.sx;
      ri.w = (int)(((long)n * p.y_img_stride + (long)oy * p.y_row_stride + (long)ox * p.y_pix_stride) * 4);
    }
    rowinfo[t] = ri;
  }
  __syncthreads();

  // loader state: this thread stages rows (t / C4) + RPP*i, float4 column (t % C4)
  const int lrow = t / C4;
  const int lc4 = (t % C4) * 4;
  unsigned avoff[NA], amask[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const i32x4 ri = rowinfo[lrow + RPP * i];
    avoff[i] = (unsigned)ri.x + lc4 * 4;
    unsigned mk = 0;
    for (int tp = 0; tp < p.ntaps; ++tp) {  // which taps fall inside the image for this pixel / column
      const int iy = ri.y + p.taps[tp].dy;
      const int xe = ri.z + lc4 + p.taps[tp].xoff;
      const bool ok = (unsigned)iy < (unsigned)p.x_rows && (unsigned)xe < (unsigned)p.x_rowlen;
      mk |= (ok ? 1u : 0u) << tp;
    }
    amask[i] = mk;
  }
  unsigned bvoff[NBV];
#pragma unroll
  for (int j = 0; j < NBV; ++j) {
    const int n = n0 + lrow + RPP * j;
    bvoff[j] = n < p.Cout ? (unsigned)(n * p.Ktot + lc4) * 4u : kOOB;  // rows past Cout read as zeros
  }
  // the source descriptor starts `x_bias` elements BEFORE the tensor so that every tap displacement is a
  // non-negative soffset; masked lanes never touch memory, valid lanes land inside the tensor
  const __amdgpu_buffer_rsrc_t xr = dc_rsrc(p.x + p.x_bias, 0x7fffffffu);
  const __amdgpu_buffer_rsrc_t wr_ = dc_rsrc(p.w, 0x7fffffffu);

  f32x16 acc[FM][FN];
#pragma unroll
  for (int a = 0; a < FM; ++a)
#pragma unroll
    for (int b = 0; b < FN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int T = p.Ktot / BK;
  int tap = 0, c0 = 0, kg = 0;  // all uniform (SALU)
  f32x4 ra[NA], rb[NBV];

  auto gload_a = [&]() {
    const ConvTap tp = p.taps[tap];
    const unsigned soff = (unsigned)(tp.soff + c0) * 4u;
    const unsigned bit = 1u << tap;
#pragma unroll
    for (int i = 0; i < NA; ++i) ra[i] = dc_bload4(xr, (amask[i] & bit) ? avoff[i] : kOOB, soff);
    c0 += BK;
    if (c0 >= tp.klen) {
      c0 = 0;
      ++tap;
    }
  };
  auto gload_b = [&]() {
#pragma unroll
    for (int j = 0; j < NBV; ++j) rb[j] = dc_bload4(wr_, bvoff[j], (unsigned)kg * 4u);
    kg += BK;
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
  // MFMA operand fragments, two register sets (chunk parity)
  f32x4 av[2][FM], bv[2][FN];
  const int frag_a = (wr * TM + (lane & 31)) * LDK + (lane >> 5) * 4 + wk * 8;
  const int frag_b = BM * LDK + (wc * TN + (lane & 31)) * LDK + (lane >> 5) * 4 + wk * 8;
  auto frag_load = [&](int buf, int q, int set) {
    const float* As = smem + buf * TILE + frag_a + q * WK * 8;
    const float* Bs = smem + buf * TILE + frag_b + q * WK * 8;
#pragma unroll
    for (int a = 0; a < FM; ++a) av[set][a] = *reinterpret_cast<const f32x4*>(As + a * 32 * LDK);
#pragma unroll
    for (int b = 0; b < FN; ++b) bv[set][b] = *reinterpret_cast<const f32x4*>(Bs + b * 32 * LDK);
  };

  // Software pipeline.  A wave issues in order, so anything that is not interleaved BETWEEN MFMAs in
  // program order is paid on top of them.  Per K-tile the wave owns NCH chunks of 4 MFMA steps; the
  // non-MFMA work of the iteration is placed into those steps and pinned with sched_barrier:
  //   chunk 0, step 0: ds_read the fragments of chunk 1
  //            step 1: ds_write tile it+1 (registers loaded one iteration ago) into the idle LDS stage
  //            step 2/3: issue the global loads of tile it+2 (A rows / filter rows)
  //   chunk q, step 0: ds_read the fragments of chunk q+1
  //   last chunk, step 1: barrier, then ds_read chunk 0 of tile it+1 from the stage just filled —
  //            its latency is covered by the 3 remaining MFMA steps of this tile.
  gload_a();
  gload_b();
  lstore(0);
  if (T > 1) {
    gload_a();
    gload_b();
  }
  __syncthreads();
  frag_load(0, 0, 0);
  for (int it = 0; it < T; ++it) {
    const int buf = it & 1;
    const bool more1 = it + 1 < T, more2 = it + 2 < T;
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
      const int cur = q & 1;
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        if (st == 0 && q + 1 < NCH) frag_load(buf, q + 1, cur ^ 1);
        if (st == 1 && q + 1 == NCH && more1) {  // after step 0 has consumed (waited for) this chunk's operands
          __syncthreads();
          frag_load(buf ^ 1, 0, 0);
        }
        if (q == 0 && st == 1 && more1) lstore(buf ^ 1);
        if (q == 0 && st == 2 && more2) gload_a();
        if (q == 0 && st == 3 && more2) gload_b();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int a = 0; a < FM; ++a)
#pragma unroll
          for (int b = 0; b < FN; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][a][st], bv[cur][b][st], acc[a][b], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  __syncthreads();

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
  const __amdgpu_buffer_rsrc_t yr = dc_rsrc(p.y, 0x7fffffffu);
  const __amdgpu_buffer_rsrc_t rr = dc_rsrc(p.resid ? p.resid : p.y, 0x7fffffffu);
#pragma unroll
  for (int b = 0; b < FN; ++b) {
    const int co = n0 + wc * TN + b * 32 + (lane & 31);
    const bool cok = co < p.Cout;
    const float sc = (cok && p.scale) ? p.scale[co] : 1.f;
    const float sh = (cok && p.shift) ? p.shift[co] : 0.f;
    const bool sig = co < p.sigmoid_ch;
#pragma unroll
    for (int a = 0; a < FM; ++a) {
      unsigned off[16];
      float rs[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int yo = rowinfo[wr * TM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)].w;
        off[r] = (yo >= 0 && cok) ? (unsigned)yo + co * 4 : kOOB;  // masked lanes: load 0 / store dropped
      }
      if (p.resid) {  // all 16 shortcut loads in flight at once
#pragma unroll
        for (int r = 0; r < 16; ++r)
          rs[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, off[r], 0, 0));
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) rs[r] = 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[a][b][r] * sc + sh + rs[r];
        if (p.relu) v = fmaxf(v, 0.f);
        if (sig) v = 1.f / (1.f + expf(-v));
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yr, off[r], 0, 0);
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
    DC_VARIANT(32, 32, 64, 1, 1, 4),    // 8: same for K segments that are only multiples of 64
    DC_VARIANT(32, 64, 32, 1, 2, 2),    // 9: K segments that are only multiples of 32 (the stem)
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

int launch_conv_gemm(const ConvGemmParams& p_in, int variant, void* stream) {
  if (variant < 0 || variant >= kNumVariants) return (int)hipErrorInvalidValue;
  const VariantEntry& e = kVariants[variant];
  ConvGemmParams p = p_in;
  if (p.ntaps < 1 || p.ntaps > 32) return (int)hipErrorInvalidValue;  // tap-validity masks are 32 bits
  int bias = 0;
  for (int i = 0; i < p.ntaps; ++i) {
    if (p.taps[i].klen % e.BK != 0) return (int)hipErrorInvalidValue;
    // a tap whose x-validity depends on the channel block (the row-tap stem) must be one K tile
    bias = std::min(bias, p.taps[i].dy * p.x_row_stride + p.taps[i].xoff);
  }
  p.x_bias = bias;
  for (int i = 0; i < p.ntaps; ++i) p.taps[i].soff = p.taps[i].dy * p.x_row_stride + p.taps[i].xoff - bias;
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
