// wino_f16.hip — Winograd F(2x2, 3x3) for the stride-1 3x3 convolutions of a float16 net (gfx950), tile name `wino_h23`.
//
// The reference runs these layers as im2col + SGEMM (src/caffe/layers/conv_layer.cpp:25-40, base_conv_layer.cpp:257-280);
// the product's direct form is the gather-GEMM of kernels.hip.  This kernel computes the same sums as
// Y = A^T [ (G g G^T) . (B^T d B) ] A per 4x4 input patch d -> 2x2 outputs: 16 independent GEMMs over the input channels
// (one per transform position (i, j)), 2.25x fewer MFMA flops than the direct form.  float16 operands, float32 accumulation
// (v_mfma_f32_32x32x16_f16), float32 epilogue — the arithmetic contract of DC_OPT_DTYPE 1.
//
// Shape of a workgroup (round 6; every step below was measured on the res4 3x3 layer at batch 8, tools/wino_f16_probe.py +
// DC_DEBUG_TIMING stamps + SQ counters):
//  * TWO tile blocks of 4 x 8 tiles (consecutive in the block list: they may lie in different rows or images) x 64 output
//    channels, 8 waves = (transform row i) x (position pair {2 jp, 2 jp + 1}).  A wave owns positions (i, 2 jp), (i, 2 jp + 1)
//    of both blocks and both 32-channel fragments: 8 accumulators of 32x32 = 128 registers.  The transformed filters are the
//    MFMA's ROW operand, the transformed patches its COLUMN operand: a lane's 16 accumulator registers are 16 output channels
//    of ONE tile, so the epilogue forms 16-byte channel vectors in registers.
//    (First form: one block x 128 channels, a wave = (row, channel half): 64 KB of filter fragments per 16-channel sub-step
//    and workgroup, 1 577 cycles per sub-step for 512 of MFMA.  Second: two blocks x 64 channels, a wave = (row, channel half)
//    with all four positions: 32 KB per sub-step, 1 295 cycles, the waves parked at s_waitcnt / s_barrier 43 % of their cycles —
//    all eight waves leave a barrier together, issue 64 reads together and wait for the last of them; no faster without the
//    filter loads, 28 % faster without the reads.  A second patch register set for reads in flight spilled, and a spill
//    reload is a vector-memory load that drains the filter fragments in flight in front of it: 31 k cycles for 21 k.  The
//    (row, position pair) split needs 24 + 12 patch / row registers instead of 32 + 16 and fits.)
//  * the 10 x 18 input pixels of each block are staged once per 32 channels in LDS (ring of 3, ONE barrier per 32 channels),
//    pre-multiplied by 1/4 — B^T d B sums four pixels, and a trunk whose activations use float16's range would overflow the
//    transformed patch; the factor is exact (a power of two) and is undone by the float32 epilogue scale;
//  * a UNIT = one tile block of one 16-channel sub-step: 6 ds_read_b128 (two patch rows x three patch columns), 12 + 8
//    packed float16 adds (B^T rows, then the two positions), 4 MFMAs.  The reads of unit u + 1 are issued in the middle of
//    unit u, into the registers the row adds have just consumed; the first unit of a stage is read at the end of the previous
//    step (stage K + 1 is complete while step K runs), so nothing waits on LDS right behind a barrier.
//    The transform arithmetic is packed float16 (v_pk_add_f16): two roundings per transformed value.  float32 intermediates
//    through v_fma_mix_f32 / v_fma_mixlo_f16 (one rounding) were built and measured: a mix instruction costs a wave ~8 cycles
//    where a packed add costs 4, the K loop ran 27 k cycles instead of 17.7 k — and both forms sit inside the float16 path's
//    stated bounds (tests/test_gpu_winograd_f16.py);
//  * LDS layout: 64 bytes per pixel (four 16-byte chunks of 8 channels, chunk q of pixel column px stored at q ^ ((px >> 2) & 3)),
//    row pitch 74 chunks, and the fragment's columns dealt to tiles so that the 16 lanes a ds_read_b128 serves together
//    ({0-3,12-15,20-27} / {4-11,16-19,28-31}, MI355X_MICROARCH.md) hold tile rows {0,1} / {2,3}: adjacent tiles are TWO pixels
//    apart, so without the swizzle and the row pairing every group would hit 8 of its 16 bank quads twice
//    (SQ_LDS_BANK_CONFLICT of the kernel: 0);
//  * the transformed filters (16/9 of the filter bytes, no reuse inside a wave) are packed on the host in fragment order
//    — one wave load = 1 KiB contiguous — and read straight from global memory into registers (32 KB per sub-step and
//    workgroup, no fragment fetched twice by a workgroup, none through LDS), the next sub-step's fragment into the registers the
//    MFMA has just read (HUB = 1; two register sets — two sub-steps ahead — spill: EXPERIMENTS.md I);
//  * inverse transform: the wave's two positions in registers, the other position pair and the four rows through LDS (128 KB:
//    the staging ring's memory and more), one tile block per round; then folded BatchNorm/Scale, ReLU, one rounding to
//    float16, v_permlane32_swap -> 16-byte stores.
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
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int HBTY = 4, HBTX = 8;                      // tiles per block
constexpr int HRH = 2 * HBTY + 2, HRW = 2 * HBTX + 2;  // staged pixels: 10 x 18
constexpr int HKC = 32;                                // channels per staged step (two 16-channel MFMA sub-steps)
constexpr int HP16 = 74;                               // 16-byte chunks per staged pixel row (18 x 4 + 2: see the layout note above)
constexpr int HCHUNKS = HRH * HRW * 4;                 // 720 chunks per block and stage
constexpr int HBLK16 = HRH * HP16;                     // chunks of one staged tile block
constexpr int HSTAGE16 = 2 * HBLK16 + 2;               // two tile blocks + the dump slot of the staging threads past them
constexpr int HNTH = 512;
constexpr int HPART16 = 8 * 2 * 2 * 4 * 64;            // partial inverse transforms of a round: [wave][b][fragment][quad][lane] x 16 bytes = 128 KB
constexpr int HLDS16 = HPART16;                        // the staging ring (3 x 23.7 KB) lives in the same memory
static_assert(3 * HSTAGE16 <= HLDS16, "the staging ring lives in the memory the partials reuse");
static_assert((HLDS16 + 32) * 16 <= 160 * 1024, "LDS of a CU");
constexpr unsigned kOOBh = 0x80000000u;
#ifndef DC_WINO_LATE_STORE
#define DC_WINO_LATE_STORE 0
#endif
#ifndef DC_WINO_ABL
#define DC_WINO_ABL 0  // diagnostics builds only (tools/wino_f16_ablate.sh): 1 no barrier in the K loop, 2 MFMAs only, 3 everything but the MFMAs
#endif
constexpr int HUB = 1;  // filter-fragment register sets: 1 = the next sub-step's fragments into the registers just read, 2 = two sub-steps ahead

__device__ __forceinline__ __amdgpu_buffer_rsrc_t h_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ u32x4 h_bload4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ int h_fastdiv(int n, const unsigned (&mg)[2]) {
  return (mg[1] >> 31) ? n : (int)(__umulhi((unsigned)n, mg[0]) >> (mg[1] & 31));
}
__device__ __forceinline__ f16x2 h_as_h2(unsigned v) { return __builtin_bit_cast(f16x2, v); }
__device__ __forceinline__ unsigned h_as_u(f16x2 v) { return __builtin_bit_cast(unsigned, v); }
// lanes 32..63 of lo <-> lanes 0..31 of hi (one s_nop: the VALU instructions that produced the operands need wait states)
__device__ __forceinline__ void h_permlane32_swap2(unsigned (&lo)[2], unsigned (&hi)[2]) {
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3" : "+v"(lo[0]), "+v"(lo[1]), "+v"(hi[0]), "+v"(hi[1]));
}
}  // namespace

__global__ __launch_bounds__(HNTH, 1) void wino_h23_kernel(const ConvGemmParams p) {
  const long long t_entry = (long long)__builtin_amdgcn_s_memrealtime();
  // the five 64-byte lines of the argument block this kernel reads are requested together at entry (as DC_KARG_TOUCH of kernels.hip:
  // fetched lazily, field by field, each first touch of a line is a scalar-cache miss on the critical path)
  unsigned ka0, ka1, ka2, ka3, ka4;
  {
    const auto* ka_ = __builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("s_load_dword %0, %5, 0x0\n\ts_load_dword %1, %5, 0x40\n\ts_load_dword %2, %5, 0x80\n\ts_load_dword %3, %5, 0xc0\n\t"
                 "s_load_dword %4, %5, 0x100"
                 : "=&s"(ka0), "=&s"(ka1), "=&s"(ka2), "=&s"(ka3), "=&s"(ka4)
                 : "s"(ka_));
  }
  static_assert(offsetof(ConvGemmParams, wide_epi) < 0x140, "the fields read below lie in the five lines touched");
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
  // ---- what depends on the lane only (no kernel argument): fragment column -> tile, so that the lanes a ds_read_b128 serves
  //      together hold tile rows {0,1} or {2,3}
  const int col = lane & 31, kg = lane >> 5;
  const bool in0 = col < 4 || (col >= 12 && col < 16) || (col >= 20 && col < 28);
  const int idx = in0 ? (col < 4 ? col : col < 16 ? col - 8 : col - 12) : (col < 12 ? col - 4 : col < 20 ? col - 8 : col - 16);
  const int ty = (in0 ? 0 : 2) + (idx >> 3), tx = idx & 7;
  const int wi = wave & 3, jp = wave >> 2;  // transform row, position pair {2 jp, 2 jp + 1}
  int rofs[3][2];  // chunk index of (patch column jp + x, sub-step h) in patch row 0 of this lane's tile
#pragma unroll
  for (int x = 0; x < 3; ++x)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int px = 2 * tx + jp + x;
      rofs[x][h] = 2 * ty * HP16 + px * 4 + ((2 * h + kg) ^ ((px >> 2) & 3));
    }
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
  asm volatile("" ::"s"(ka0), "s"(ka1), "s"(ka2), "s"(ka3), "s"(ka4));  // (the dummy loads have landed: their registers are free)
  // filter fragments: [Cout/32][4 i][C/16][4 j][64 lanes][8 halves]; this wave reads positions (wi, 2 jp + {0, 1}) of the two
  // fragments nb * 2 + {0, 1}: the fragment stride travels in the scalar offset.  Requested first: they need no pixel decode
  const __amdgpu_buffer_rsrc_t ur = h_rsrc(p.w);
  const unsigned uvo = ((unsigned)((nb * 2 * 4 + wi) * (C / 16)) * 4096u) + (unsigned)(2 * jp) * 1024u + (unsigned)lane * 16u;
  const unsigned ufrag = (unsigned)(4 * (C / 16)) * 4096u;
  u32x4 ub[HUB][2][2];  // [sub-step parity (HUB = 2)][position jj][fragment]
  auto bload = [&](int s, int jj, int nf, int k16) { ub[s][jj][nf] = h_bload4(ur, uvo + (unsigned)jj * 1024u, (unsigned)k16 * 4096u + (unsigned)nf * ufrag); };
  const int NS = C / HKC;
#pragma unroll
  for (int s = 0; s < HUB; ++s)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int nf = 0; nf < 2; ++nf) bload(s, jj, nf, s);
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
  // ---- staging: thread t moves chunks t, t + 512, t + 1024 of the 2 x 720 (pixel, 8-channel chunk) pairs of a stage
  const __amdgpu_buffer_rsrc_t xr = h_rsrc(p.x);
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
    // (the whole tensor stays below 2 GiB: wino_half_eligible — the image base travels in the 32-bit offset)
    gofs[q] = ok ? (unsigned)((long)bn[m] * p.x_img_stride + iy * p.x_row_stride + ix * C + cq * 8) * 2u : kOOBh;
    sofs[q] = e0 < 2 * HCHUNKS ? m * HBLK16 + py * HP16 + px * 4 + (cq ^ ((px >> 2) & 3)) : 2 * HBLK16 + (t & 1);
  }
  u32x4 g[3], g1[3];
  auto gload = [&](int K) {
#pragma unroll
    for (int q = 0; q < 3; ++q) g[q] = h_bload4(xr, gofs[q], (unsigned)(K * HKC * 2));
  };
  const f16x2 quarter = {(_Float16)0.25f, (_Float16)0.25f};
  auto sstore = [&](int buf, const u32x4 (&gg)[3]) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      u32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = h_as_u(h_as_h2(gg[q][e]) * quarter);
      lds[buf * HSTAGE16 + sofs[q]] = v;
    }
  };
  stamp(1);
  gload(0);
  if (NS > 1) {
#pragma unroll
    for (int q = 0; q < 3; ++q) g1[q] = h_bload4(xr, gofs[q], (unsigned)(HKC * 2));
  }
  if (t < 128) {  // the epilogue's per-channel constants of the workgroup's 64 channels wait in LDS
    const float* src = t < 64 ? p.scale : p.shift;
    reinterpret_cast<float*>(&lds[HLDS16])[t] = src ? src[nb * 64 + (t & 63)] : (t < 64 ? 1.f : 0.f);
  }
  f32x16 acc[2][2][2];  // [tile block][position 2 jp + jj][32-channel fragment]
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int nf = 0; nf < 2; ++nf)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][jj][nf][r] = 0.f;
  stamp(2);
  sstore(0, g);
  if (NS > 2) gload(2);
  if (NS > 1) sstore(1, g1);
  stamp(3);

  // everything from here on depends on the wave's role only through compile-time constants: the patch rows and columns it reads
  // become immediate offsets of the ds_reads, the signs of B^T become add / subtract
  auto body = [&](auto i_tag, auto jp_tag) {
    constexpr int I = decltype(i_tag)::value, JP = decltype(jp_tag)::value;
    // B^T row I as a combination of two patch rows: 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3
    constexpr int RA = I == 0 ? 0 : (I == 2 ? 2 : 1), RB = I == 0 ? 2 : (I == 1 ? 2 : (I == 2 ? 1 : 3));
    constexpr int SB = I == 1 ? 1 : -1;
    // positions 0, 1 need the row's columns 0..2 (V0 = t0 - t2, V1 = t1 + t2), positions 2, 3 columns 1..3 (V2 = t2 - t1, V3 = t1 - t3)
    struct Patch {
      u32x4 a[3], b[3];  // the two patch rows of a unit: [column JP + x] x 8 channels
    };
    struct Rows {
      f16x2 t[3][4];  // [column JP + x][dword]: B^T row I of the patch, packed float16
    };
    auto rd = [&](auto u_tag, auto h_tag, auto m_tag, Patch& pt) {
      constexpr int U = decltype(u_tag)::value, HH = decltype(h_tag)::value, M = decltype(m_tag)::value;
#if DC_WINO_ABL == 2
      return;
#endif
      __builtin_amdgcn_sched_barrier(0);  // (the scheduler would hoist the reads above the adds that free their registers)
#pragma unroll
      for (int x = 0; x < 3; ++x) {
        pt.a[x] = lds[U * HSTAGE16 + M * HBLK16 + RA * HP16 + rofs[x][HH]];
        pt.b[x] = lds[U * HSTAGE16 + M * HBLK16 + RB * HP16 + rofs[x][HH]];
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    auto stage1 = [&](const Patch& pt, Rows& r) {
#if DC_WINO_ABL == 2
      return;
#endif
#pragma unroll
      for (int x = 0; x < 3; ++x)
#pragma unroll
        for (int e = 0; e < 4; ++e) r.t[x][e] = SB > 0 ? h_as_h2(pt.a[x][e]) + h_as_h2(pt.b[x][e]) : h_as_h2(pt.a[x][e]) - h_as_h2(pt.b[x][e]);
    };
    // S: parity of the unit's sub-step (which fragment register set); LOAD: the unit's MFMAs are the last readers of that set —
    // each requests the fragment of sub-step k16 (HUB sub-steps ahead) into the registers it has just read
    auto stage2 = [&](const Rows& r, auto s_tag, auto m_tag, auto load_tag, int k16) {
      constexpr int S = decltype(s_tag)::value % HUB, M = decltype(m_tag)::value;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        // local columns: JP = 0: V0 = x0 - x2, V1 = x1 + x2;  JP = 1: V2 = x1 - x0, V3 = x0 - x2
        constexpr int XA[2][2] = {{0, 1}, {1, 0}}, XB[2][2] = {{2, 2}, {0, 2}};
        const int xa = XA[JP][jj], xb = XB[JP][jj];
        const bool plus = JP == 0 && jj == 1;
        u32x4 v;
#if DC_WINO_ABL == 2
        v = u32x4{(unsigned)rofs[jj][0], (unsigned)rofs[jj][1], (unsigned)lane, (unsigned)r.t[0][0][0]};
#else
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = h_as_u(plus ? r.t[xa][e] + r.t[xb][e] : r.t[xa][e] - r.t[xb][e]);
#endif
#pragma unroll
        for (int nf = 0; nf < 2; ++nf) {
#if DC_WINO_ABL == 3
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[M][jj][nf][e] += __builtin_bit_cast(float, ub[S][jj][nf][e] ^ v[e]);
#else
          acc[M][jj][nf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ub[S][jj][nf]), __builtin_bit_cast(f16x8, v), acc[M][jj][nf], 0, 0, 0);
#endif
          if constexpr (decltype(load_tag)::value) bload(S, jj, nf, k16);
        }
      }
    };
    using T0 = std::integral_constant<int, 0>;
    using T1 = std::integral_constant<int, 1>;
    Patch pt;
    Rows r;
    auto step = [&](int K, auto u_tag, auto un_tag) {
      constexpr int U = decltype(u_tag)::value;
      const int last = 2 * NS - 1;  // (loads past the end re-read the last sub-step: harmless, never used)
#if DC_WINO_ABL != 1
      __syncthreads();  // stages <= K+1 are complete; buffer (K+2) % 3 is free
#endif
      stage1(pt, r), rd(u_tag, T0{}, T1{}, pt), stage2(r, T0{}, T0{}, T0{}, 0);
      stage1(pt, r), rd(u_tag, T1{}, T0{}, pt), stage2(r, T0{}, T1{}, T1{}, min(2 * K + HUB, last));
#if !DC_WINO_LATE_STORE
      if (K + 2 < NS) sstore((U + 2) % 3, g);
#endif
      stage1(pt, r), rd(u_tag, T1{}, T1{}, pt), stage2(r, T1{}, T0{}, T0{}, 0);
      stage1(pt, r);
      if (K + 1 < NS) rd(un_tag, T0{}, T0{}, pt);  // the first unit of the next stage (complete since this step's barrier)
      stage2(r, T1{}, T1{}, T1{}, min(2 * K + 1 + HUB, last));
#if DC_WINO_LATE_STORE
      if (K + 2 < NS) sstore((U + 2) % 3, g);
#endif
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
    // ---- inverse transform: the wave's two positions in registers (P[b] = sum over its j of M[I][j] A[j][b]; A^T = [1 1 1 0; 0 1 -1 -1]),
    //      the other position pair and the four rows through LDS, one tile block per round; in a round this wave finalises output
    //      pixel (a, bq) = (I >> 1, I & 1) of every tile for the channel quads 2 JP, 2 JP + 1 of both fragments
    constexpr int A = I >> 1, BQ = I & 1;
    const __amdgpu_buffer_rsrc_t yr = h_rsrc(p.y);
    __syncthreads();  // every wave is done reading the staging ring, which the partials now overwrite
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
      for (int nf = 0; nf < 2; ++nf)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          f32x4 p0, p1;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int rr = rq * 4 + e;
            if constexpr (JP == 0) p0[e] = acc[m][0][nf][rr] + acc[m][1][nf][rr], p1[e] = acc[m][1][nf][rr];
            else p0[e] = acc[m][0][nf][rr], p1[e] = -acc[m][0][nf][rr] - acc[m][1][nf][rr];
          }
          lds[((((JP * 4 + I) * 2 + 0) * 2 + nf) * 4 + rq) * 64 + lane] = __builtin_bit_cast(u32x4, p0);
          lds[((((JP * 4 + I) * 2 + 1) * 2 + nf) * 4 + rq) * 64 + lane] = __builtin_bit_cast(u32x4, p1);
        }
      const int oy = bphy[m] + d * (2 * (bby[m] * HBTY + ty) + A), ox = bphx[m] + d * (2 * (bbx[m] * HBTX + tx) + BQ);
      const unsigned ybase = bok[m] && oy < p.OH && ox < p.OW ? (unsigned)((long)bn[m] * p.y_img_stride + oy * p.y_row_stride + ox * p.y_pix_stride) * 2u : kOOBh;
      __syncthreads();
      if (m == 0) stamp(5), stamp(6);
#pragma unroll
      for (int nf = 0; nf < 2; ++nf) {
        unsigned hv[2][2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int rq = 2 * JP + k;
          f32x4 q[4];
#pragma unroll
          for (int i2 = 0; i2 < 4; ++i2)
            if (i2 != (A == 0 ? 3 : 0))
              q[i2] = __builtin_bit_cast(f32x4, lds[((((0 * 4 + i2) * 2 + BQ) * 2 + nf) * 4 + rq) * 64 + lane]) +
                      __builtin_bit_cast(f32x4, lds[((((1 * 4 + i2) * 2 + BQ) * 2 + nf) * 4 + rq) * 64 + lane]);
          f32x4 y = A == 0 ? q[0] + q[1] + q[2] : q[1] - q[2] - q[3];
          // D layout: row = 8 * (r / 4) + 4 * (lane / 32) + r % 4 -> channel nf * 32 + rq * 8 + kg * 4 + e of the workgroup's 64
          const f32x4 sc = __builtin_bit_cast(f32x4, lds[HLDS16 + nf * 8 + rq * 2 + kg]);
          const f32x4 sh = __builtin_bit_cast(f32x4, lds[HLDS16 + 16 + nf * 8 + rq * 2 + kg]);
          y = y * sc + sh;
          if (p.relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = fmaxf(y[e], 0.f);
          }
          hv[k][0] = h_as_u(f16x2{(_Float16)y[0], (_Float16)y[1]});
          hv[k][1] = h_as_u(f16x2{(_Float16)y[2], (_Float16)y[3]});
        }
        // lanes 0..31 hold channels 8 rq + 0..3 of their tile, lanes 32..63 channels 8 rq + 4..7: after the swap of the quads
        // (2 JP, 2 JP + 1) lanes 0..31 hold channels 16 JP + 0..7 and lanes 32..63 channels 16 JP + 8..15 — one 16-byte store
        h_permlane32_swap2(hv[0], hv[1]);
        const u32x4 o = {hv[0][0], hv[0][1], hv[1][0], hv[1][1]};
        __builtin_amdgcn_raw_buffer_store_b128(o, yr, ybase + (unsigned)(nb * 64 + nf * 32 + 16 * JP + 8 * kg) * 2u, 0, 0);
      }
      if (m == 0) __syncthreads();  // the next round's partials overwrite what this round still reads
    }
    stamp(7);
  };
  using C0 = std::integral_constant<int, 0>;
  using C1 = std::integral_constant<int, 1>;
  if (wi == 0) jp ? body(std::integral_constant<int, 0>{}, C1{}) : body(std::integral_constant<int, 0>{}, C0{});
  else if (wi == 1) jp ? body(std::integral_constant<int, 1>{}, C1{}) : body(std::integral_constant<int, 1>{}, C0{});
  else if (wi == 2) jp ? body(std::integral_constant<int, 2>{}, C1{}) : body(std::integral_constant<int, 2>{}, C0{});
  else jp ? body(std::integral_constant<int, 3>{}, C1{}) : body(std::integral_constant<int, 3>{}, C0{});
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
  hipLaunchKernelGGL(wino_h23_kernel, gd, bd, 0, (hipStream_t)stream, q);
  return (int)hipGetLastError();
}

}  // namespace dc
