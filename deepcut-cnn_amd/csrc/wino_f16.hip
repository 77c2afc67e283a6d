// wino_f16.hip — Winograd F(2x2, 3x3) for the stride-1 3x3 convolutions of a float16 net (gfx950).
//
// The reference runs these layers as im2col + SGEMM (src/caffe/layers/conv_layer.cpp:25-40, base_conv_layer.cpp:257-280);
// the product's direct form is the gather-GEMM of kernels.hip.  This kernel computes the same sums as
// Y = A^T [ (G g G^T) . (B^T d B) ] A per 4x4 input patch d -> 2x2 outputs: 16 independent GEMMs over the input channels
// (one per transform position (i, j)), 2.25x fewer MFMA flops than the direct form.  float16 operands, float32 accumulation
// (v_mfma_f32_32x32x16_f16), float32 epilogue — the arithmetic contract of DC_OPT_DTYPE 1.
//
// What shapes it (round 6; the float32 sibling is wino_f23_kernel in kernels.hip, whose geometry it shares):
//  * the f16 matrix pipe is 16x the f32 one, so a workgroup must amortise the input transform over many output channels:
//    workgroup = 4 x 8 tiles (ONE 32-column MFMA fragment) x 64*NF output channels, 8 waves = (transform row i) x (channel
//    half); a wave owns the positions (i, 0..3) of NF 32-channel fragments: 4*NF accumulators of 32x32 (128 registers at
//    NF = 2).  The transformed filters are the MFMA's ROW operand, the transformed patches its COLUMN operand: a lane's 16
//    accumulator registers are 16 output channels of ONE tile, so the epilogue forms 16-byte channel vectors in registers;
//  * the 10 x 18 input pixels of the block are staged once per 32 channels in LDS (ring of 3, ONE barrier per 32 channels),
//    pre-multiplied by 1/4 — B^T d B sums four pixels, and a trunk whose activations use float16's range would overflow
//    the transformed patch; the factor is exact (a power of two) and is undone by the float32 epilogue scale;
//  * a wave reads the two patch rows its transform row needs (8 ds_read_b128 per 16 channels) and transforms them in
//    registers right in front of its MFMAs: float32 intermediates through v_fma_mix_f32 / v_fma_mixlo_f16 (conversion folded
//    into the add, ONE rounding to float16 per transformed value), or, TM = 0, packed float16 adds (two roundings);
//  * LDS layout: 64 bytes per pixel (four 16-byte chunks of 8 channels, chunk q of pixel column px stored at q ^ ((px >> 2) & 3)),
//    row pitch 74 chunks, and the fragment's columns dealt to tiles so that the 16 lanes a ds_read_b128 serves together
//    ({0-3,12-15,20-27} / {4-11,16-19,28-31}, MI355X_MICROARCH.md) hold tile rows {0,1} / {2,3}: adjacent tiles are TWO pixels
//    apart, so without the swizzle and the row pairing every group would hit 8 of its 16 bank quads twice;
//  * the transformed filters (16/9 of the filter bytes, no reuse inside a wave) are packed on the host in fragment order
//    — one wave load = 1 KiB contiguous — and read straight from global memory into registers, one 16-channel sub-step
//    ahead, into the registers the previous sub-step's MFMAs have just consumed (no second register set: the wave stays at
//    two per SIMD);
//  * inverse transform: over j in registers, over i (four waves) through LDS (the staging ring's memory), then folded
//    BatchNorm/Scale, ReLU, one rounding to float16, v_permlane32_swap -> 16-byte stores.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "kernels.h"

namespace dc {

namespace {
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int HBTY = 4, HBTX = 8;                      // tiles per block
constexpr int HRH = 2 * HBTY + 2, HRW = 2 * HBTX + 2;  // staged pixels: 10 x 18
constexpr int HKC = 32;                                // channels per staged step (two 16-channel MFMA sub-steps)
constexpr int HP16 = 74;                               // 16-byte chunks per staged pixel row (18 x 4 + 2: see the layout note above)
constexpr int HCHUNKS = HRH * HRW * 4;                 // 720 chunks per stage
constexpr int HBLK16 = HRH * HP16;                     // chunks of one staged tile block
constexpr int HSTAGE16 = 2 * HBLK16 + 2;               // two tile blocks + the dump slot of the staging threads past them
constexpr int HNTH = 512;
constexpr int HLDS16 = 3 * HSTAGE16;                   // 71 KB: the ring; the partial inverse transforms of a round (64 KB) reuse it
static_assert(HLDS16 >= 4096, "the partials of one round (8 waves x 2 x 4 x 64 x 16 bytes) live in the staging ring's memory");
constexpr unsigned kOOBh = 0x80000000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t h_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ u32x4 h_bload4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ int h_fastdiv(int n, const unsigned (&mg)[2]) {
  return (mg[1] >> 31) ? n : (int)(__umulhi((unsigned)n, mg[0]) >> (mg[1] & 31));
}
// float32 <- (float16 half HI of a) + S * (float16 half HI of b): conversion folded into the add (v_fma_mix_f32), exact
template <int HI, int S>
__device__ __forceinline__ float h_mix_add(unsigned a, unsigned b) {
  float d;
  if constexpr (HI == 0) {
    if constexpr (S > 0) asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(b), "v"(a));
    else asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(b), "v"(a));
  } else {
    if constexpr (S > 0) asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(b), "v"(a));
    else asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(b), "v"(a));
  }
  return d;
}
// packed float16 pair <- round(x0 + S * y0), round(x1 + S * y1): ONE rounding each (v_fma_mixlo_f16 / v_fma_mixhi_f16)
template <int S>
__device__ __forceinline__ unsigned h_mix_pack(float x0, float y0, float x1, float y1) {
  unsigned d;
  if constexpr (S > 0) {
    asm("v_fma_mixlo_f16 %0, %1, 1.0, %2" : "=v"(d) : "v"(y0), "v"(x0));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, %2" : "+v"(d) : "v"(y1), "v"(x1));
  } else {
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2" : "=v"(d) : "v"(y0), "v"(x0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2" : "+v"(d) : "v"(y1), "v"(x1));
  }
  return d;
}
__device__ __forceinline__ f16x2 h_as_h2(unsigned v) { return __builtin_bit_cast(f16x2, v); }
__device__ __forceinline__ unsigned h_as_u(f16x2 v) { return __builtin_bit_cast(unsigned, v); }
// lanes 32..63 of lo <-> lanes 0..31 of hi (one s_nop: the VALU instructions that produced the operands need wait states)
__device__ __forceinline__ void h_permlane32_swap2(unsigned (&lo)[2], unsigned (&hi)[2]) {
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3" : "+v"(lo[0]), "+v"(lo[1]), "+v"(hi[0]), "+v"(hi[1]));
}
}  // namespace

// TM: 1 = float32 transform intermediates (one rounding of every transformed value), 0 = packed float16 adds (two roundings)
//
// Round-6 measurements that set the workgroup's shape (tools/wino_f16_probe.py, res4 3x3 at batch 8): the first form — ONE tile
// block x 128 channels per workgroup, every wave streaming the fragments of two 32-channel slices — ran its K loop at 1 577 cycles
// per 16-channel sub-step for 512 cycles of MFMA: 64 KB of filter fragments per sub-step through a vector-memory path that
// delivers ~45 B/clk to a CU (240 workgroups x 1 MB = 252 MB per launch: the L2s' rate).  A filter fragment must therefore
// serve more tiles: workgroup = TWO tile blocks (consecutive in the block list: they may lie in different rows or images) x 64
// output channels, a wave = (transform row i) x (32-channel half) and BOTH blocks with the same fragment registers — 32 KB of
// fragments per sub-step, no fragment fetched twice by a workgroup, none through LDS.
template <int TM>
__global__ __launch_bounds__(HNTH, 1) void wino_h23_kernel(const ConvGemmParams p) {
  const long long t_entry = (long long)__builtin_amdgcn_s_memrealtime();
  __shared__ __attribute__((aligned(16))) u32x4 lds[HLDS16 + 32];  // the ring / the partials, then 64 scales + 64 shifts
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  auto stamp = [&](int slot) {  // DC_DEBUG_TIMING: per-wave phase stamps (format of conv_gemm_kernel)
    if (p.dbg && lane == 0) {
      long long* dd = p.dbg + ((long)blockIdx.x * 8 + wave) * 12;
      dd[slot] = (long long)__builtin_readcyclecounter();
      if (slot == 0) dd[8] = t_entry, dd[10] = (long long)__builtin_amdgcn_s_memrealtime();
      if (slot == 7) dd[9] = (long long)__builtin_amdgcn_s_memrealtime();
    }
  };
  stamp(0);
  const int C = p.klen, H = p.x_rows, W = p.x_rowlen / p.klen;
  const int d = p.ddy;  // dilation: d*d interleaved phase images, each an ordinary pad-1 problem on the pixels (phy + d*u, phx + d*v)
  const int NBY = p.w_NBY, NBX = p.w_NBX, nblk = p.w_nblk, npair = (nblk + 1) >> 1;
  // workgroup b runs on XCD (b % 8): every XCD takes a contiguous range of the (channel-block-major) logical grid, so that the
  // filter fragments of a channel block are pulled through one L2 (a locality hint: any bijection computes the same result)
  int lb = blockIdx.x;
  if (p.xcd_on) {
    const int g8 = gridDim.x >> 3, r8 = gridDim.x & 7, q = blockIdx.x & 7;
    lb = q * g8 + min(q, r8) + (blockIdx.x >> 3);
  }
  const int nb = h_fastdiv(lb, p.w_div_nblk), pair = lb - nb * npair;  // (w_div_nblk divides by the number of block PAIRS here)
  const int wi = wave & 3, nh = wave >> 2;
  // the two tile blocks of the workgroup
  int bn[2], bphy[2], bphx[2], bby[2], bbx[2];
  bool bok[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int blk = 2 * pair + m;
    bok[m] = blk < nblk;
    const int nph = h_fastdiv(blk, p.w_div_nbyx), brem = blk - nph * (NBY * NBX);
    bn[m] = h_fastdiv(nph, p.w_div_dd);
    const int ph = nph - bn[m] * (d * d);
    bphy[m] = h_fastdiv(ph, p.w_div_d), bphx[m] = ph - bphy[m] * d;
    bby[m] = h_fastdiv(brem, p.w_div_nbx), bbx[m] = brem - bby[m] * NBX;
  }
  if (t < 128) {  // the epilogue's per-channel constants of the workgroup's 64 channels wait in LDS
    const float* src = t < 64 ? p.scale : p.shift;
    reinterpret_cast<float*>(&lds[HLDS16])[t] = src ? src[nb * 64 + (t & 63)] : (t < 64 ? 1.f : 0.f);
  }

  // ---- staging: thread t moves chunks t, t + 512, t + 1024 of the 2 x 720 (pixel, 8-channel chunk) pairs of a stage
  const __amdgpu_buffer_rsrc_t xr = h_rsrc(p.x);
  const __amdgpu_buffer_rsrc_t ur = h_rsrc(p.w);
  unsigned gofs[3];
  int sofs[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int e0 = t + q * HNTH;
    const int m = e0 >= HCHUNKS ? 1 : 0, e = e0 - m * HCHUNKS;
    const int pix = e >> 2, cq = e & 3;
    const int py = pix / HRW, px = pix - py * HRW;
    const int oy0 = 2 * HBTY * bby[m] - 1, ox0 = 2 * HBTX * bbx[m] - 1;  // phase-grid coordinates of staged pixel (0, 0): pad 1
    const int iy = bphy[m] + d * (oy0 + py), ix = bphx[m] + d * (ox0 + px);
    const bool ok = e0 < 2 * HCHUNKS && bok[m] && oy0 + py >= 0 && ox0 + px >= 0 && iy < H && ix < W;
    // (one image stays below 2 GiB: wino_half_eligible; the image base travels in the offset as long as the whole tensor does too)
    gofs[q] = ok ? (unsigned)((long)bn[m] * p.x_img_stride + iy * p.x_row_stride + ix * C + cq * 8) * 2u : kOOBh;
    sofs[q] = e0 < 2 * HCHUNKS ? m * HBLK16 + py * HP16 + px * 4 + (cq ^ ((px >> 2) & 3)) : 2 * HBLK16 + (t & 1);
  }
  // ---- fragment column -> tile: the lanes a ds_read_b128 serves together hold tile rows {0,1} or {2,3}
  const int col = lane & 31, kg = lane >> 5;
  const bool in0 = col < 4 || (col >= 12 && col < 16) || (col >= 20 && col < 28);
  const int idx = in0 ? (col < 4 ? col : col < 16 ? col - 8 : col - 12) : (col < 12 ? col - 4 : col < 20 ? col - 8 : col - 16);
  const int ty = (in0 ? 0 : 2) + (idx >> 3), tx = idx & 7;
  int rofs[4][2];  // chunk index of (patch column c, sub-step h) in patch row 0 of this lane's tile
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int px = 2 * tx + c;
      rofs[c][h] = 2 * ty * HP16 + px * 4 + ((2 * h + kg) ^ ((px >> 2) & 3));
    }
  // filter fragments: [Cout/32][4 i][C/16][4 j][64 lanes][8 halves]; this wave's fragment is nb * 2 + nh
  const unsigned uvo = ((unsigned)(((nb * 2 + nh) * 4 + wi) * (C / 16)) * 4096u) + (unsigned)lane * 16u;

  f32x16 acc[2][4];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][j][r] = 0.f;
  u32x4 g[3], ub[4];
  const f16x2 quarter = {(_Float16)0.25f, (_Float16)0.25f};
  auto gload = [&](int K) {
#pragma unroll
    for (int q = 0; q < 3; ++q) g[q] = h_bload4(xr, gofs[q], (unsigned)(K * HKC * 2));
  };
  auto sstore = [&](int buf, const u32x4 (&gg)[3]) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      u32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = h_as_u(h_as_h2(gg[q][e]) * quarter);
      lds[buf * HSTAGE16 + sofs[q]] = v;
    }
  };
  auto bload = [&](int j, int k16) { ub[j] = h_bload4(ur, uvo + (unsigned)j * 1024u, (unsigned)k16 * 4096u); };
  const int NS = C / HKC;
  stamp(1);
  u32x4 g1[3];
  gload(0);
#pragma unroll
  for (int j = 0; j < 4; ++j) bload(j, 0);
  if (NS > 1) {
#pragma unroll
    for (int q = 0; q < 3; ++q) g1[q] = h_bload4(xr, gofs[q], (unsigned)(HKC * 2));
  }
  stamp(2);
  sstore(0, g);
  if (NS > 2) gload(2);
  if (NS > 1) sstore(1, g1);
  stamp(3);

  // everything from here on depends on the wave's transform row only through compile-time constants: the patch rows it
  // reads become immediate offsets of the ds_reads, the signs of B^T become add / subtract
  // Software pipeline of a wave (a UNIT = one tile block of one 16-channel sub-step: 8 ds_read_b128, 32 packed adds, 4 MFMAs): the
  // reads of unit u + 1 are issued in the middle of unit u (see stage1 / stage2 below).  Measured before this
  // (res4 3x3, batch 8): 1 295 cycles per sub-step for 512 of MFMA, the waves parked at s_waitcnt / s_barrier 43 % of their cycles
  // (SQ_WAIT_ANY) — all eight waves of the workgroup leave a barrier together, issue 64 reads together and wait for the last of
  // them —, no faster without the filter loads, 28 % faster without the reads.  The first unit of a stage is read at the end of the
  // previous step (stage K + 1 is complete while step K runs), so nothing waits on LDS right behind a barrier.
  auto body = [&](auto i_tag) {
    constexpr int I = decltype(i_tag)::value;
    // B^T row I as a combination of two patch rows: 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3
    constexpr int RA = I == 0 ? 0 : (I == 2 ? 2 : 1), RB = I == 0 ? 2 : (I == 1 ? 2 : (I == 2 ? 1 : 3));
    constexpr int SB = I == 1 ? 1 : -1;
    struct Patch {
      u32x4 a[4], b[4];  // the two patch rows of a unit: [patch column] x 8 channels
    };
    auto rd = [&](auto u_tag, auto h_tag, auto m_tag, Patch& d) {
      constexpr int U = decltype(u_tag)::value, HH = decltype(h_tag)::value, M = decltype(m_tag)::value;
      __builtin_amdgcn_sched_barrier(0);  // (the scheduler would hoist the reads above the adds that free their registers)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        d.a[c] = lds[U * HSTAGE16 + M * HBLK16 + RA * HP16 + rofs[c][HH]];
        d.b[c] = lds[U * HSTAGE16 + M * HBLK16 + RB * HP16 + rofs[c][HH]];
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    // A unit in program order: stage 1 of the transform (B^T rows: 16 packed adds) consumes the patch registers; the reads of
    // the NEXT unit are issued into those same registers right behind it; then stage 2 (4 packed adds per position) and the MFMA of
    // each position j in turn — the reads' latency runs under stage 2 and the MFMAs, and no second patch register set is needed
    // (a full second set spilled 98 registers).  LOAD: the unit's MFMAs are the last readers of the filter fragments — each
    // requests the fragment of sub-step k16 into the registers it has just read.
    struct Rows {
      f16x2 t[4][4];  // TM = 0: [patch column][dword] packed float16
      float tl[4][4], th[4][4];  // TM = 1: the low / high half of each packed pair, float32
    };
    auto stage1 = [&](const Patch& d, Rows& r) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if constexpr (TM == 1) {
            r.tl[c][e] = h_mix_add<0, SB>(d.a[c][e], d.b[c][e]);
            r.th[c][e] = h_mix_add<1, SB>(d.a[c][e], d.b[c][e]);
          } else {
            r.t[c][e] = SB > 0 ? h_as_h2(d.a[c][e]) + h_as_h2(d.b[c][e]) : h_as_h2(d.a[c][e]) - h_as_h2(d.b[c][e]);
          }
        }
    };
    auto stage2 = [&](const Rows& r, auto m_tag, auto load_tag, int k16) {
      constexpr int M = decltype(m_tag)::value;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        constexpr int XA[4] = {0, 1, 2, 1}, XB[4] = {2, 2, 1, 3};  // V_j = t[XA[j]] -/+ t[XB[j]]  (+ for j = 1)
        u32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if constexpr (TM == 1)
            v[e] = j == 1 ? h_mix_pack<1>(r.tl[XA[j]][e], r.tl[XB[j]][e], r.th[XA[j]][e], r.th[XB[j]][e])
                          : h_mix_pack<-1>(r.tl[XA[j]][e], r.tl[XB[j]][e], r.th[XA[j]][e], r.th[XB[j]][e]);
          else
            v[e] = h_as_u(j == 1 ? r.t[XA[j]][e] + r.t[XB[j]][e] : r.t[XA[j]][e] - r.t[XB[j]][e]);
        }
        acc[M][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ub[j]), __builtin_bit_cast(f16x8, v), acc[M][j], 0, 0, 0);
        if constexpr (decltype(load_tag)::value) bload(j, k16);
      }
    };
    using T0 = std::integral_constant<int, 0>;
    using T1 = std::integral_constant<int, 1>;
    Patch pt;
    Rows r;
    auto step = [&](int K, auto u_tag, auto un_tag) {
      constexpr int U = decltype(u_tag)::value;
      const int k16_tail = 2 * K + 2 < 2 * NS ? 2 * K + 2 : 0;  // (the tail load is a harmless re-read of step 0)
      __syncthreads();  // stages <= K+1 are complete; buffer (K+2) % 3 is free
      stage1(pt, r), rd(u_tag, T0{}, T1{}, pt), stage2(r, T0{}, T0{}, 0);
      stage1(pt, r), rd(u_tag, T1{}, T0{}, pt), stage2(r, T1{}, T1{}, 2 * K + 1);
      if (K + 2 < NS) sstore((U + 2) % 3, g);
      stage1(pt, r), rd(u_tag, T1{}, T1{}, pt), stage2(r, T0{}, T0{}, 0);
      stage1(pt, r);
      if (K + 1 < NS) rd(un_tag, T0{}, T0{}, pt);  // the first unit of the next stage (complete since this step's barrier)
      stage2(r, T1{}, T1{}, k16_tail);
      if (K + 3 < NS) gload(K + 3);
    };
    __syncthreads();  // stages 0 and 1 are in LDS
    rd(std::integral_constant<int, 0>{}, T0{}, T0{}, pt);
    for (int K0 = 0; K0 < NS; K0 += 3) {
      step(K0, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
      if (K0 + 1 < NS) step(K0 + 1, std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{});
      if (K0 + 2 < NS) step(K0 + 2, std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{});
    }
    stamp(4);
    // ---- inverse transform: over j in registers (P[b] = sum_j M[I][j] A[j][b]), over i (four waves) through LDS, one tile
    //      block per round; this wave then finalises output pixel (a, bq) = (I >> 1, I & 1) of every tile of the block
    constexpr int A = I >> 1, BQ = I & 1;
    const __amdgpu_buffer_rsrc_t yr = h_rsrc(p.y);
    const int co0 = (nb * 2 + nh) * 32;  // first channel of this wave's fragment
    __syncthreads();  // every wave is done reading the staging ring, which the partials now overwrite
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        f32x4 p0, p1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = rq * 4 + e;
          p0[e] = acc[m][0][r] + acc[m][1][r] + acc[m][2][r];
          p1[e] = acc[m][1][r] - acc[m][2][r] - acc[m][3][r];
        }
        lds[(((nh * 4 + I) * 2 + 0) * 4 + rq) * 64 + lane] = __builtin_bit_cast(u32x4, p0);
        lds[(((nh * 4 + I) * 2 + 1) * 4 + rq) * 64 + lane] = __builtin_bit_cast(u32x4, p1);
      }
      __syncthreads();
      if (m == 0) stamp(5), stamp(6);
      const int oy = bphy[m] + d * (2 * (bby[m] * HBTY + ty) + A), ox = bphx[m] + d * (2 * (bbx[m] * HBTX + tx) + BQ);
      const unsigned ybase = bok[m] && oy < p.OH && ox < p.OW ? (unsigned)((long)bn[m] * p.y_img_stride + oy * p.y_row_stride + ox * p.y_pix_stride) * 2u : kOOBh;
      unsigned hv[4][2];
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const f32x4 q0 = __builtin_bit_cast(f32x4, lds[(((nh * 4 + 0) * 2 + BQ) * 4 + rq) * 64 + lane]);
        const f32x4 q1 = __builtin_bit_cast(f32x4, lds[(((nh * 4 + 1) * 2 + BQ) * 4 + rq) * 64 + lane]);
        const f32x4 q2 = __builtin_bit_cast(f32x4, lds[(((nh * 4 + 2) * 2 + BQ) * 4 + rq) * 64 + lane]);
        const f32x4 q3 = __builtin_bit_cast(f32x4, lds[(((nh * 4 + 3) * 2 + BQ) * 4 + rq) * 64 + lane]);
        f32x4 y = A == 0 ? q0 + q1 + q2 : q1 - q2 - q3;
        // D layout: row = 8 * (r / 4) + 4 * (lane / 32) + r % 4 -> channel nh * 32 + rq * 8 + kg * 4 + e of the workgroup's 64
        const f32x4 sc = __builtin_bit_cast(f32x4, lds[HLDS16 + nh * 8 + rq * 2 + kg]);
        const f32x4 sh = __builtin_bit_cast(f32x4, lds[HLDS16 + 16 + nh * 8 + rq * 2 + kg]);
        y = y * sc + sh;
        if (p.relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) y[e] = fmaxf(y[e], 0.f);
        }
        hv[rq][0] = h_as_u(f16x2{(_Float16)y[0], (_Float16)y[1]});
        hv[rq][1] = h_as_u(f16x2{(_Float16)y[2], (_Float16)y[3]});
      }
      // lanes 0..31 hold channels 8 rq + 0..3 of their tile, lanes 32..63 channels 8 rq + 4..7: after the swap of the pairs
      // (rq, rq + 1) lanes 0..31 hold channels 16 k + 0..7 and lanes 32..63 channels 16 k + 8..15 — 16-byte stores
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        h_permlane32_swap2(hv[2 * k], hv[2 * k + 1]);
        const u32x4 o = {hv[2 * k][0], hv[2 * k][1], hv[2 * k + 1][0], hv[2 * k + 1][1]};
        __builtin_amdgcn_raw_buffer_store_b128(o, yr, ybase + (unsigned)(co0 + 16 * k + 8 * kg) * 2u, 0, 0);
      }
      if (m == 0) __syncthreads();  // the next round's partials overwrite what this round still reads
    }
    stamp(7);
  };
  if (wi == 0) body(std::integral_constant<int, 0>{});
  else if (wi == 1) body(std::integral_constant<int, 1>{});
  else if (wi == 2) body(std::integral_constant<int, 2>{});
  else body(std::integral_constant<int, 3>{});
}

// ---- host side ------------------------------------------------------------------------------------------------------------
bool wino_half_eligible(const ConvGemmParams& p) {
  const int d = p.ddy;  // dilation (1 or more), the same along x and y, with pad = dilation ("same" convolution)
  if (p.esize != 2 || p.nty != 3 || p.ntx != 3 || p.sy != 1 || d < 1 || d > 4 || p.dy0 != -d) return false;
  const int C = p.klen;
  if (C <= 0 || C % HKC != 0 || p.Cout % 64 != 0 || p.sigmoid_ch != 0 || p.ncls > 1) return false;
  if (p.sx != C || p.ddx != d * C || p.x0 != -d * C) return false;        // stride 1, dilation d, pad d along x
  if (p.x_rowlen % C != 0 || p.x_row_stride != p.x_rowlen) return false;  // dense NHWC rows of C channels
  if (p.OH != p.x_rows || p.OW != p.x_rowlen / C) return false;           // "same" convolution
  if (p.y_pix_stride % 8 != 0 || p.y_row_stride % 8 != 0 || p.y_img_stride % 8 != 0) return false;  // 16-byte output vectors
  // 32-bit byte offsets (buffer addressing): the input, the output and the packed filter image stay below 2 GiB each
  if ((long long)p.NB * p.x_img_stride * 2 >= 0x7fffffffLL || (long long)p.NB * p.y_img_stride * 2 >= 0x7fffffffLL ||
      (long long)wino_half_packed_elems(p.Cout, C) * 2 >= 0x7fffffffLL)
    return false;
  return true;
}

long wino_half_grid(const ConvGemmParams& p) {
  const int d = p.ddy;
  const int TY = ((p.OH + d - 1) / d + 1) / 2, TX = ((p.OW + d - 1) / d + 1) / 2;
  const long nblk = (long)p.NB * d * d * ((TY + HBTY - 1) / HBTY) * ((TX + HBTX - 1) / HBTX);
  return (nblk + 1) / 2 * (p.Cout / 64);  // a workgroup = two consecutive tile blocks x 64 output channels
}

size_t wino_half_packed_elems(int Cout, int Cin) { return (size_t)16 * Cout * Cin; }

// g: [Cout][Cin][3][3] (Caffe order) -> U = G g G^T per (co, ci) in double, each output channel multiplied by the exact power
// of two that brings its largest |U| into [2^13, 2^14) (row_scale[co] = its inverse: DevVec::row_scale), laid out in MFMA
// fragment order [Cout/32][4 i][Cin/16][4 j][64 lanes][8]: lane = 32 * ((ci % 16) / 8) + co % 32, element = ci % 8
void wino_half_pack_filters(const float* g, int Cout, int Cin, bool rowscale, float* out, float* row_scale) {
  static const double G[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
  std::vector<double> U((size_t)Cin * 16);
  for (int co = 0; co < Cout; ++co) {
    double mx = 0;
    for (int ci = 0; ci < Cin; ++ci) {
      const float* w = g + ((size_t)co * Cin + ci) * 9;
      double tmp[4][3];
      for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 3; ++b) tmp[a][b] = G[a][0] * w[b] + G[a][1] * w[3 + b] + G[a][2] * w[6 + b];
      for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b) {
          const double u = tmp[a][0] * G[b][0] + tmp[a][1] * G[b][1] + tmp[a][2] * G[b][2];
          U[(size_t)ci * 16 + a * 4 + b] = u;
          mx = std::max(mx, std::fabs(u));
        }
    }
    double f = 1.0;
    if (rowscale && mx > 0 && std::isfinite(mx)) {
      int k = 13 - std::ilogb(mx);
      k = std::max(-60, std::min(60, k));
      f = std::ldexp(1.0, k);
    }
    row_scale[co] = (float)(1.0 / f);
    const int fr = co / 32, row = co % 32;
    for (int ci = 0; ci < Cin; ++ci) {
      const int k16 = ci / 16, kq = (ci % 16) / 8, e = ci % 8;
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
          out[(((((size_t)fr * 4 + i) * (Cin / 16) + k16) * 4 + j) * 64 + kq * 32 + row) * 8 + e] = (float)(U[(size_t)ci * 16 + i * 4 + j] * f);
    }
  }
}

// DC_WINO_HALF_PK=1: packed float16 transform adds (two roundings per transformed value) instead of float32 intermediates — the
// A/B switch of tests/test_gpu_winograd_f16.py; read at every launch (launches are captured into graphs: not a hot path)
static int wino_half_tm() {
  const char* e = getenv("DC_WINO_HALF_PK");
  return e && atoi(e) ? 0 : 1;
}

int launch_wino_half(const ConvGemmParams& p, void* stream) {
  if (!wino_half_eligible(p) || p.resid) return (int)hipErrorInvalidValue;
  const long grid = wino_half_grid(p);
  if (grid <= 0) return 0;
  if (grid > 0x7fffffffL) return (int)hipErrorInvalidValue;
  ConvGemmParams q = p;
  static const int xcd_map = getenv("DC_XCD_MAP") ? atoi(getenv("DC_XCD_MAP")) : 1;
  q.xcd_on = xcd_map && grid >= 16;
  auto magic = [](unsigned dv, unsigned (&mg)[2]) {  // dc_magic of kernels.hip: sh = 31 + ceil(log2 dv), mul = floor(2^sh / dv) + 1
    if (dv <= 1) {
      mg[0] = 0, mg[1] = 0x80000000u;
      return;
    }
    int l = 0;
    while ((1ull << l) < dv) ++l;
    const int sh = 31 + l;
    const unsigned long long qq = (unsigned long long)((((unsigned __int128)1) << sh) / dv);
    mg[0] = (unsigned)(qq + 1), mg[1] = (unsigned)(sh - 32);
  };
  {
    const int d = p.ddy;
    q.w_TY = ((p.OH + d - 1) / d + 1) / 2, q.w_TX = ((p.OW + d - 1) / d + 1) / 2;
    q.w_NBY = (q.w_TY + HBTY - 1) / HBTY, q.w_NBX = (q.w_TX + HBTX - 1) / HBTX;
    q.w_nblk = p.NB * d * d * q.w_NBY * q.w_NBX;
    magic((unsigned)((q.w_nblk + 1) / 2), q.w_div_nblk);  // (the kernel divides by the number of block PAIRS)
    magic((unsigned)(q.w_NBY * q.w_NBX), q.w_div_nbyx);
    magic((unsigned)(d * d), q.w_div_dd);
    magic((unsigned)d, q.w_div_d);
    magic((unsigned)q.w_NBX, q.w_div_nbx);
  }
  const dim3 gd((unsigned)grid), bd(HNTH);
  if (wino_half_tm() == 1) hipLaunchKernelGGL((wino_h23_kernel<1>), gd, bd, 0, (hipStream_t)stream, q);
  else hipLaunchKernelGGL((wino_h23_kernel<0>), gd, bd, 0, (hipStream_t)stream, q);
  return (int)hipGetLastError();
}

}  // namespace dc
