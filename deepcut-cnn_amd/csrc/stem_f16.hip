// stem_f16.hip — the 7x7 / stride-2 stem convolution of a float16 net (ResNet-152.prototxt conv1: 3 -> 64 channels, pad 3, + BatchNorm/Scale +
// ReLU) as a kernel of its own ("stem7x7").
//
// Why: the gather-GEMM sees the stem as 7 row-taps of 64 elements (a float16 image has 8 channels per pixel, 3 of them real: 7 x 8 = 56
// elements of a kernel row, padded to the float16 K granule): every output pixel pulls 7 x 128 bytes through LDS-DMA — 717 MB of L2 -> LDS
// traffic at batch 8 for a 51 MB image — and two thirds of the matrix instructions multiply zeros: 79.6-90.6 us at batch 8 (1.4 TB/s of useful
// bytes, 164 TFLOP/s), 300 us of the four-scale pyramid.
// Here a workgroup stages the band of the image its 8 x 64 outputs read ONCE (21 rows x 134 pixels, the first 4 channels of each: 22 KB;
// per-pixel bounds = the zero padding), keeps the filters in registers in MFMA operand order (7 rows x 2 K-steps of 16: the 28 elements
// kx 4 + ci of a row + 4 zeros, 112 registers for the 64 channels), and a wave walks 32-pixel row segments: 14 fragment reads (two adjacent
// 4-channel pixels = one 16-byte chunk, always aligned because the band starts 3 pixels left of an even pixel) + 28 matrix instructions
// per 32 x 64 outputs: 35.0 us at batch 8, 9.7 at batch 1 (16.7) — profiles/r06_stem_probe.txt.  The
// epilogue is the one of stream1x1.hip (affine in fp32, ReLU, 16-byte vectors by v_permlane32_swap, transposed in a wave-private LDS tile
// to whole 128-byte pixel rows).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "kernels.h"

namespace dc {
namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr unsigned kOOBt = 0x80000000u;
#pragma clang diagnostic ignored "-Winline-asm"

constexpr int TRH = 8, TCW = 64;                 // conv rows x conv columns of a workgroup's tile
constexpr int BROWS = 2 * TRH + 5;               // image rows of its band
constexpr int BPX = 2 * TCW + 6;                 // image pixels of a band row: 2 (TCW - 1) + 7, + 1 behind the zero-weight eighth tap
constexpr int BROWB = (BPX * 8 + 15) / 16 * 16;  // bytes of a band row (a pixel = 4 halves)
constexpr int BANDB = BROWS * BROWB;
constexpr int SOBUF = 32 * 128;                  // a wave's output tile: 32 pixels x 64 channels of halves

struct StemArgs {
  const void* x;  // [NB][H][W][x_pix] halves, the first 4 of a pixel are read (3 channels + a zero)
  const void* w;  // stem_pack_filters image
  const float* scale;
  const float* shift;
  void* y;
  int NB, H, W, OH, OW;
  int x_row, x_pix, y_row, y_pix;   // elements between image rows / image pixels / output rows / output pixels
  long x_img, y_img;                // elements between images
  int relu, tiles_x, tiles_y;
  unsigned div_tx[2], div_txy[2];
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t t_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ int t_fastdiv(int n, const unsigned (&mg)[2]) {
  return (mg[1] >> 31) ? n : (int)(__umulhi((unsigned)n, mg[0]) >> (mg[1] & 31));
}
__device__ __forceinline__ void t_permlane32_swap4(float (&lo)[4], float (&hi)[4]) {
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %4\n\tv_permlane32_swap_b32 %1, %5\n\tv_permlane32_swap_b32 %2, %6\n\tv_permlane32_swap_b32 %3, %7"
      : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]));
}

template <bool RELU>
__global__ __launch_bounds__(256, 2) void stem7x7_kernel(const StemArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[BANDB + 4 * SOBUF + 2 * 64 * 4];
  float* scl = reinterpret_cast<float*>(smem + BANDB + 4 * SOBUF);
  float* shl = scl + 64;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int p32 = lane & 31, h = lane >> 5;
  // tile: image n, conv rows [r0, r0 + TRH), conv columns [c0, c0 + TCW)
  const int bx = blockIdx.x;
  const int n = t_fastdiv(bx, a.div_txy), rem = bx - n * a.tiles_x * a.tiles_y;
  const int ty = t_fastdiv(rem, a.div_tx), tx = rem - ty * a.tiles_x;
  const int r0 = ty * TRH, c0 = tx * TCW;

  // ---- filters: registers (A operand: rows = channels), [fragment][kernel row][K step]
  u32x4 wreg[2][7][2];
  {
    const __amdgpu_buffer_rsrc_t wr = t_rsrc(a.w);
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int ky = 0; ky < 7; ++ky)
#pragma unroll
        for (int s = 0; s < 2; ++s)
          wreg[f][ky][s] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, (unsigned)lane * 16u, (unsigned)(((f * 7 + ky) * 2 + s) * 1024), 0));
  }
  if (t < 64) {
    scl[t] = a.scale ? a.scale[t] : 1.f;
    shl[t] = a.shift ? a.shift[t] : 0.f;
  }
  // ---- the band: image rows 2 r0 - 3 ..., pixels 2 c0 - 3 ...; a pixel (8 bytes) per request, out of the image = 0
  {
    const __amdgpu_buffer_rsrc_t xr = t_rsrc(reinterpret_cast<const _Float16*>(a.x) + (long)n * a.x_img);
    constexpr int NPX = BROWS * BPX, NIT = (NPX + 255) / 256;
    const int iy0 = 2 * r0 - 3, ix0 = 2 * c0 - 3;
    // every request of the thread first, then the LDS writes: one memory round trip for the band instead of one per pixel
    u32x2 v[NIT];
    int dst[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int q = t + i * 256;
      const int br = q / BPX, bp = q - br * BPX;
      const int iy = iy0 + br, ix = ix0 + bp;
      const unsigned off = (q < NPX && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) ? (unsigned)(iy * a.x_row + ix * a.x_pix) * 2u : kOOBt;
      v[i] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(xr, off, 0, 0));
      dst[i] = q < NPX ? br * BROWB + bp * 8 : -1;
    }
#pragma unroll
    for (int i = 0; i < NIT; ++i)
      if (dst[i] >= 0) *reinterpret_cast<u32x2*>(smem + dst[i]) = v[i];
  }
  __syncthreads();

  const __amdgpu_buffer_rsrc_t yr = t_rsrc(reinterpret_cast<_Float16*>(a.y) + (long)n * a.y_img);
  unsigned char* const obp = smem + BANDB + wave * SOBUF;
  const int oswz = (p32 >> 1) & 7;
  const f16x2 zero2 = {(_Float16)0.f, (_Float16)0.f};
  // tasks of the tile: (row, 32-column block), dealt to the waves in turn
  for (int task = wave; task < TRH * (TCW / 32); task += 4) {
    const int rr = task >> 1, cb = task & 1;
    const int orow = r0 + rr, ocol0 = c0 + cb * 32;
    if (orow >= a.OH || ocol0 >= a.OW) continue;  // (uniform per wave)
    f32x16 acc[2];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;
    // lane (pixel p32, K half h), kernel row ky, K step s: image pixels 2 (ocol0 + p32) - 3 + 4 s + 2 h and the next one = band pixels
    // 2 (cb 32 + p32) + 4 s + 2 h, + 1: one aligned 16-byte chunk
    const unsigned char* bp0 = smem + (2 * rr) * BROWB + (2 * (cb * 32 + p32) + 2 * h) * 8;
    u32x4 xf[14];
#pragma unroll
    for (int ky = 0; ky < 7; ++ky)
#pragma unroll
      for (int s = 0; s < 2; ++s) xf[ky * 2 + s] = *reinterpret_cast<const u32x4*>(bp0 + ky * BROWB + s * 32);
#pragma unroll
    for (int ky = 0; ky < 7; ++ky)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int f = 0; f < 2; ++f)
          acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wreg[f][ky][s]), __builtin_bit_cast(f16x8, xf[ky * 2 + s]), acc[f], 0, 0, 0);
    // epilogue (stream1x1.hip): register r of fragment f is channel f 32 + 8 (r >> 2) + 4 h + (r & 3) of pixel p32
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int cb0 = f * 32 + 16 * j + 4 * h, cb1 = cb0 + 8;
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(scl + cb0), h0 = *reinterpret_cast<const f32x4*>(shl + cb0);
        const f32x4 s1 = *reinterpret_cast<const f32x4*>(scl + cb1), h1 = *reinterpret_cast<const f32x4*>(shl + cb1);
        float lo[4], hi[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          lo[e] = acc[f][8 * j + e] * s0[e] + h0[e];
          hi[e] = acc[f][8 * j + 4 + e] * s1[e] + h1[e];
        }
        t_permlane32_swap4(lo, hi);
        u32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float x0 = i < 2 ? lo[2 * i] : hi[2 * i - 4], x1 = i < 2 ? lo[2 * i + 1] : hi[2 * i - 3];
          const f32x2 xp = {x0, x1};
          f16x2 hp = __builtin_convertvector(xp, f16x2);
          if (RELU) hp = __builtin_elementwise_max(hp, zero2);
          o[i] = __builtin_bit_cast(unsigned, hp);
        }
        *reinterpret_cast<u32x4*>(obp + p32 * 128 + (((f * 4 + j * 2 + h) ^ oswz) * 16)) = o;
      }
    // whole pixel rows of 128 bytes: 8 lanes each
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int px = i * 8 + lane / 8, sl = lane % 8;
      const u32x4 v = *reinterpret_cast<const u32x4*>(obp + i * 1024 + lane * 16);
      const int oc = ocol0 + px;
      const unsigned off = oc < a.OW ? (unsigned)(orow * a.y_row + oc * a.y_pix) * 2u + (unsigned)((sl ^ ((px >> 1) & 7)) * 16) : kOOBt;
      __builtin_amdgcn_raw_buffer_store_b128(v, yr, off, 0, 0);
    }
  }
}

void t_magic(unsigned dv, unsigned (&mg)[2]) {  // dc_magic of kernels.hip
  if (dv <= 1) {
    mg[0] = 0, mg[1] = 0x80000000u;
    return;
  }
  int l = 0;
  while ((1ull << l) < dv) ++l;
  const int sh = 31 + l;
  const unsigned long long qq = (unsigned long long)((((unsigned __int128)1) << sh) / dv);
  mg[0] = (unsigned)(qq + 1), mg[1] = (unsigned)(sh - 32);
}
}  // namespace

// the launch the lowering makes of the stem: 7 row-taps of one 64-element K segment over an image of CP = 4 or 8 channels per pixel
// (at most 4 of them real: the caller's business), stride 2, pad 3
bool stem7x7_eligible(const ConvGemmParams& p) {
  if (p.esize != 2 || p.ncls > 1 || p.nprob > 0 || p.sigmoid_ch != 0 || p.resid) return false;
  const int CP = p.sx / 2;
  if ((CP != 4 && CP != 8) || p.sx != 2 * CP || p.x0 != -3 * CP) return false;
  if (p.nty != 7 || p.ntx != 1 || p.klen != 64 || p.Ktot != 7 * 64 || p.dy0 != -3 || p.ddy != 1 || p.sy != 2) return false;
  if (p.Cout != 64 || p.x_rowlen % CP != 0 || p.x_row_stride < p.x_rowlen) return false;
  const int W = p.x_rowlen / CP, H = p.x_rows;
  if (p.OH != (H + 6 - 7) / 2 + 1 || p.OW != (W + 6 - 7) / 2 + 1) return false;
  if ((p.y_pix_stride * 2) % 16 != 0 || (p.y_row_stride * 2) % 16 != 0 || (p.y_img_stride * 2) % 16 != 0) return false;
  if ((long)p.x_rows * p.x_row_stride * 2 >= 0x7fffffffL || (long)p.OH * p.y_row_stride * 2 >= 0x7fffffffL) return false;  // 32-bit offsets inside an image
  return true;
}

long stem7x7_grid(const ConvGemmParams& p) { return (long)p.NB * ((p.OH + TRH - 1) / TRH) * ((p.OW + TCW - 1) / TCW); }

size_t stem7x7_packed_elems() { return (size_t)2 * 7 * 2 * 64 * 8; }

// g: [64][C][7][7] (Caffe order, C <= 4) -> [fragment 2][ky 7][K step 2][lane 64][8]: element e = kx * 4 + ci of kernel row ky (zeros for
// ci >= C and for the eighth tap) sits at lane 32 * ((e % 16) / 8) + co % 32, position e % 8 of K step e / 16
void stem7x7_pack_filters(const float* g, int C, float* out) {
  for (size_t i = 0; i < stem7x7_packed_elems(); ++i) out[i] = 0.f;
  for (int co = 0; co < 64; ++co)
    for (int ci = 0; ci < C && ci < 4; ++ci)
      for (int ky = 0; ky < 7; ++ky)
        for (int kx = 0; kx < 7; ++kx) {
          const int e = kx * 4 + ci;
          out[((((size_t)(co / 32) * 7 + ky) * 2 + e / 16) * 64 + ((e % 16) / 8) * 32 + co % 32) * 8 + e % 8] = g[(((size_t)co * C + ci) * 7 + ky) * 7 + kx];
        }
}

int launch_stem7x7(const ConvGemmParams& p, void* stream) {
  if (!stem7x7_eligible(p) || ((uintptr_t)p.x & 7) || ((uintptr_t)p.y & 15) || ((uintptr_t)p.w & 15)) return (int)hipErrorInvalidValue;
  const long grid = stem7x7_grid(p);
  if (grid <= 0) return 0;
  if (grid > 0x7fffffffL) return (int)hipErrorInvalidValue;
  StemArgs a{};
  a.x = p.x, a.w = p.w, a.scale = p.scale, a.shift = p.shift, a.y = p.y;
  a.x_pix = p.sx / 2;
  a.NB = p.NB, a.H = p.x_rows, a.W = p.x_rowlen / a.x_pix, a.OH = p.OH, a.OW = p.OW;
  a.x_row = p.x_row_stride, a.y_row = p.y_row_stride, a.y_pix = p.y_pix_stride;
  a.x_img = p.x_img_stride, a.y_img = p.y_img_stride;
  a.relu = p.relu;
  a.tiles_x = (p.OW + TCW - 1) / TCW, a.tiles_y = (p.OH + TRH - 1) / TRH;
  t_magic((unsigned)a.tiles_x, a.div_tx);
  t_magic((unsigned)(a.tiles_x * a.tiles_y), a.div_txy);
  if (p.relu) hipLaunchKernelGGL(stem7x7_kernel<true>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(stem7x7_kernel<false>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

}  // namespace dc
