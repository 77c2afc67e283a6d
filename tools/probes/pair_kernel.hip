// pair_kernel.hip — two layers of consecutive bottleneck blocks as ONE launch (float16 operands, round 5):
//
//     Y = relu( (B · W2cᵀ) ∘ a1 + b1 + S )        res<i>_branch2c + BatchNorm + Scale, + shortcut, ReLU    (1x1, WD -> 4·WD)
//     Z = relu( (Y · W2aᵀ) ∘ a2 + b2 )            res<i+1>_branch2a + BatchNorm + Scale + ReLU             (1x1, 4·WD -> WD)
//
// B is the output of branch2b (M pixels x WD channels, dense NHWC rows), S the block's input (M x 4·WD), Y the block's output —
// stored, the next block's shortcut needs it — and Z the next block's branch2a output.  The reference runs them as two SGEMMs
// per image with BatchNorm / Scale / Eltwise / ReLU passes in between (src/caffe/layers/base_conv_layer.cpp:326-341,
// batch_norm_layer.cpp:86-149, scale_layer.cpp:109-134, eltwise_layer.cpp:59-65, relu_layer.cpp:15-18); rounds 1-4 as two
// gather-GEMM launches, the second re-reading the 4·WD-wide Y (25.6 MB per res4 block at float16 batch 8) that the first had
// just written.
//
// Here Y never comes back from memory.  A wave owns 32 pixels.  The 4·WD intermediate channels are walked in CHUNKS of 64:
//   GEMM 1 (swapped operands: filters are the MFMA's row operand)  Sacc[64 ch x 32 px] = W2c[chunk] · Bᵀ      K = WD
//   epilogue 1   affine, + shortcut, ReLU, round to float16; the chunk of Y is stored; the SAME registers, packed, ...
//   GEMM 2       Zacc[WD ch x 32 px] += W2a[:, chunk] · Y[chunk]ᵀ                                                K = 64
// ... are GEMM 2's column operand: with the filters as the row operand a lane's 16 accumulators are 16 CHANNELS of one pixel,
// which is exactly the shape of v_mfma_f32_32x32x16_f16's B operand (8 K-values of one column per lane) up to a permutation of
// K — and K is a summation index, so the permutation is applied to W2a's K order ON THE HOST when the filter image is packed
// (pair_pack_filters).  No LDS round trip, no cross-lane traffic for the intermediate.  Resident per wave: its 32 rows of B as
// MFMA operands (WD/4 registers), Zacc (WD/2 registers); streamed through LDS by LDS-DMA (buffer_load ... lds, 1 KiB per wave
// instruction): the filter chunks, pre-swizzled on the host into the exact LDS image (so the requests are linear), in a ring of
// three stages, and the shortcut chunk (128 px x 128 B, double buffered).
//
// One workgroup = 4 waves = 128 pixels (one wave per SIMD: the kernel needs ~330 registers).  A launch may cover several
// tensors (the scales of a pyramid): PairArgs::prob[].
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <type_traits>

#include "pair_kernel.h"

namespace dc {

typedef float pf32x4 __attribute__((ext_vector_type(4)));
typedef float pf32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 pf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 pf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 pf16x2 __attribute__((ext_vector_type(2)));
typedef int pi32x4 __attribute__((ext_vector_type(4)));
typedef unsigned pu32x4 __attribute__((ext_vector_type(4)));
typedef unsigned pu32x2 __attribute__((ext_vector_type(2)));

constexpr unsigned kPairOOB = 0x80000000u;  // beyond any buffer: loads return 0, LDS-DMA writes 0, stores are dropped

#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ pi32x4 pair_rsrc_words(const void* p, unsigned bytes) {
  const unsigned long long a = (unsigned long long)p;
  return pi32x4{(int)(unsigned)a, (int)((a >> 32) & 0xffffu), (int)bytes, 0x00020000};
}
// LDS-DMA: 64 lanes x 16 bytes from global memory straight into LDS at M0 + 16*lane.  Inline asm: through the builtin the
// compiler would make every later ds_read wait for vmcnt(0); the kernel counts vmcnt itself (kernels.hip, dc_dma16).
__device__ __forceinline__ void pair_dma16(pi32x4 rs, unsigned lds, unsigned voff, unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds), "v"(voff), "s"(rs), "s"(soff)
               : "memory", "m0");
}
template <int N>
__device__ __forceinline__ void pair_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void pair_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// PAIR_FENCE (below): nothing is scheduled across it — inline asm orders memory operations only, and the MFMAs of one period
// would otherwise drift into the next one

// float32 + the low / high half of a packed float16 pair, rounded once (v_fma_mix_f32 h * 1.0 + f): conversion folded into the add
__device__ __forceinline__ float pair_add_half_lo(unsigned h2, float f) {
  float d;
  asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h2), "v"(f));
  return d;
}
__device__ __forceinline__ float pair_add_half_hi(unsigned h2, float f) {
  float d;
  asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h2), "v"(f));
  return d;
}
__device__ __forceinline__ unsigned pair_relu_pk(unsigned h2) {  // max(x, 0) on both halves (after rounding: monotone, 0 is exact)
  unsigned d;
  asm("v_pk_max_f16 %0, %1, 0" : "=v"(d) : "v"(h2));
  return d;
}

// Geometry of the filter image (shared by the kernel and the host packer).  Stage 2c holds chunk c of W2c: 64 rows (the chunk's
// channels) of WD halves; stage 2c+1 holds chunk c of W2a: WD rows (output channels) of 64 halves in the permuted K order.
// 16-byte pieces of a row are XOR-swizzled so that the 16 lanes a ds_read_b128 serves together hit 16 distinct bank groups.
__host__ __device__ constexpr int pair_stage_bytes(int WD) { return 128 * WD; }
__host__ __device__ inline int pair_w1_off(int WD, int row, int q) {  // byte offset inside a W2c stage: row 0..63, 16-byte piece q of WD/8
  return row * WD * 2 + ((q & ~15) | ((q & 15) ^ (row & 15))) * 16;
}
__host__ __device__ inline int pair_w2_off(int row, int q) {  // inside a W2a stage: row 0..WD-1 of 128 bytes, piece q of 8
  return row * 128 + (q ^ ((row >> 1) & 7)) * 16;
}
// the channel (within the chunk) that sits in K slot t (0..7) of lane half h in GEMM 2's step s4 (0..3): accumulator register
// i = 8*(s4&1) + t of fragment f = s4 >> 1 is channel 32f + 8*(i>>2) + 4h + (i&3)   (C layout of the 32x32 MFMAs)
__host__ __device__ inline int pair_chan(int s4, int h, int t) {
  const int i = 8 * (s4 & 1) + t;
  return 32 * (s4 >> 1) + 8 * (i >> 2) + 4 * h + (i & 3);
}

// ORDER OF WORK.  Eight waves, two per SIMD: wave (pg, r) = pixel group pg (32 of the workgroup's 128 pixels) x role r.  The two
// roles of a pixel group hold the same 32 rows of B and split the work of every chunk:
//     GEMM 1      role r computes fragment r of the chunk (its 32 channels): WD/16 MFMAs
//     epilogue 1  ... and finishes those 32 channels: affine + shortcut + ReLU, rounded to float16.  The values go back INTO the
//                 shortcut buffer in place (Y has the shortcut's shape) and leave the workgroup one period later as whole 128-byte
//                 rows (16-byte vectors, 8 lanes per row: the mirror image of the LDS-DMA that brought the shortcut), and they
//                 stay in registers, packed, as two of GEMM 2's four K steps; the other two come from the partner through a
//                 16 KB exchange buffer (written behind one barrier, read behind the next)
//     GEMM 2      role r owns output channels [r*WD/2, (r+1)*WD/2): 4 K steps x WD/64 fragments
// With one wave per SIMD (the first form of this kernel, EXPERIMENTS.md) every LDS-DMA request, store and epilogue instruction
// stalled the matrix pipe: 4 600 cycles per chunk for 2 048 of MFMA.  Two waves per SIMD issue independently, so one wave's
// memory and vector work hides under its partner's MFMAs — which needs <= 256 registers per wave, hence the split.
// The first layer's epilogue is skewed by one chunk so that it runs beside the second layer's MFMAs of the chunk before:
//     t = 0          GEMM 1 (chunk 0), epilogue 1 (0)
//     t = 2c + 1     [X(c) -> exchange buffer; rows of Y(c) -> memory]  GEMM 1 (chunk c+1)         filter stage W2c[c+1]
//     t = 2c + 2     [partner's half of X(c) <- exchange buffer]        GEMM 2 (c) || epilogue 1 (c+1)   stage W2a[c], shortcut c+1
//     t = 2*NCH - 1  [X, Y of the last chunk]  t = 2*NCH: GEMM 2 (NCH-1)
// The filter image holds the stages in this order of consumption; stage t sits in ring slot t mod 3 and is requested two
// periods ahead, its requests spread over the MFMA steps of the period.  One s_barrier per period.
#define PAIR_FENCE() __builtin_amdgcn_sched_barrier(0)
#ifndef PAIR_ABL  // diagnostic builds (WRONG results, timing only): bit 0 no in-loop LDS-DMA requests, 1 no epilogue beside GEMM 2,
#define PAIR_ABL 0  // 2 no stores of Y, 3 no exchange of the intermediate, 4 no in-loop fragment reads (operands = whatever is in the registers)
#endif
template <int WD, bool DBG = false>
__global__ __launch_bounds__(512, 1) void pair_gemm_kernel(const PairArgs ka) {
  constexpr int NC = 4 * WD, NCH = NC / 64, SB = 128 * WD, NWI = SB / 8192, KS1 = WD / 16, G2H = WD / 64;
  constexpr int RING = 0, SC = 3 * SB, XB = SC + 2 * 16384, CONSTS = XB + 16384, CONSTS2 = CONSTS + NC * 8;
  constexpr int LA = 4;   // filter fragments read ahead of the MFMA that uses them
  constexpr int DF = G2H; // MFMAs of a period held back (their fragments are in registers) until the NEXT period's first reads are
                          // in flight: every wave leaves the barrier with matrix work in hand instead of waiting out the LDS latency
  constexpr int NVM = NWI + 2;  // what every period's wait leaves in flight: the youngest stage's requests + 2 (see the order below)
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int pg = wave & 3, role = wave >> 2;
  const int px = lane & 31, h = lane >> 5;
  // DBG: shader-clock stamps of wave 0 of every workgroup -> ka.dbg[block][slot] (tools/probes/pair_probe --stamps)
  int dbg_n = 0;
  auto stamp = [&]() {
    if constexpr (DBG) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      if (wave == 0 && lane == 0 && dbg_n < 256) ka.dbg[(size_t)blockIdx.x * 256 + dbg_n] = (long long)t;
      ++dbg_n;
    }
  };

  // which tensor, which 128-pixel tile
  int k = 0;
  const int tile = blockIdx.x;
  for (int i = 1; i < kMaxPairProblems; ++i)
    if (i < ka.nprob && tile >= ka.prob[i].tile0) k = i;
  const PairProblem pr = ka.prob[k];
  const int m0 = (tile - pr.tile0) * 128;
  const int M = pr.M;
  const int row = m0 + pg * 32 + px;

  // ---- epilogue constants -> LDS (a1[NC], b1[NC]; a2[WD], b2[WD]); this pixel group's 32 rows of B -> registers
  {
    const pf32x4* src = reinterpret_cast<const pf32x4*>(ka.ab1);
    pf32x4* dst = reinterpret_cast<pf32x4*>(lds + CONSTS);
    for (int i = threadIdx.x; i < 2 * NC / 4; i += 512) dst[i] = src[i];
    const pf32x4* src2 = reinterpret_cast<const pf32x4*>(ka.ab2);
    pf32x4* dst2 = reinterpret_cast<pf32x4*>(lds + CONSTS2);
    for (int i = threadIdx.x; i < 2 * WD / 4; i += 512) dst2[i] = src2[i];
  }
  pf16x8 breg[KS1];
  {
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(pr.b), 0, (unsigned)M * WD * 2, 0x00020000);
    const unsigned vo = (unsigned)row * (WD * 2) + h * 16;
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks)
      breg[ks] = __builtin_bit_cast(pf16x8, __builtin_amdgcn_raw_buffer_load_b128(rb, row < M ? vo + ks * 32 : kPairOOB, 0, 0));
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) asm volatile("" ::"v"(breg[ks]));  // the compiler's own vmcnt waits happen HERE, not in the loop
  }
  pair_wait_vm<0>();
  __syncthreads();

  // ---- LDS-DMA requests
  const pi32x4 rw = pair_rsrc_words(ka.w, (unsigned)(2 * NCH) * SB);
  const pi32x4 rs = pair_rsrc_words(pr.s, (unsigned)M * NC * 2);
  const pi32x4 ryw = pair_rsrc_words(pr.y, (unsigned)M * NC * 2);
  const unsigned lane16 = lane * 16;
  // this wave's two row pieces of the workgroup's 128 x 128-byte shortcut / Y chunk: rows 32*pg + 16*role + 8i + lane/8, 16-byte
  // piece (lane%8) ^ swizzle(row) — the same per-lane offsets serve the LDS-DMA of the shortcut and the stores of Y
  unsigned sc_vo[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int rl = 32 * pg + 16 * role + 8 * i + (lane >> 3);
    const int piece = (lane & 7) ^ ((rl >> 1) & 7);
    sc_vo[i] = (m0 + rl < M) ? (unsigned)(m0 + rl) * (NC * 2) + piece * 16 : kPairOOB;
  }
  const unsigned sc_row0 = (32 * pg + 16 * role) * 128;
  // stage image `stage` -> ring slot `slot`: piece i of this wave's NWI (of the workgroup's 8*NWI) requests
  auto dma_w_piece = [&](int stage, int slot, int i) {
    const unsigned piece = (unsigned)(wave + 8 * i) * 1024u;
    pair_dma16(rw, RING + slot * SB + piece, stage < 2 * NCH ? lane16 : kPairOOB, (unsigned)stage * SB + piece);
  };
  auto dma_w = [&](int stage, int slot) {
#pragma unroll
    for (int i = 0; i < NWI; ++i) dma_w_piece(stage, slot, i);
  };
  auto dma_sc_piece = [&](int c, int i) {  // shortcut chunk c (128 bytes of every pixel row) -> buffer c & 1
    pair_dma16(rs, SC + (c & 1) * 16384 + sc_row0 + i * 1024, c < NCH ? sc_vo[i] : kPairOOB, (unsigned)c * 128u);
  };
  auto dma_sc = [&](int c) {
    dma_sc_piece(c, 0);
    dma_sc_piece(c, 1);
  };
  // rows of Y chunk c (finished in place in the shortcut buffer by BOTH roles: call behind a barrier) -> memory, 16 bytes per lane
  auto store_y = [&](int c) {
    const unsigned char* sb = lds + SC + (c & 1) * 16384 + sc_row0 + lane16;
    const pu32x4 v0 = *reinterpret_cast<const pu32x4*>(sb), v1 = *reinterpret_cast<const pu32x4*>(sb + 1024);
    asm volatile("s_nop 0\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen" ::"v"(v0), "v"(sc_vo[0]), "s"(ryw), "s"((unsigned)c * 128u) : "memory");
    asm volatile("s_nop 0\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen" ::"v"(v1), "v"(sc_vo[1]), "s"(ryw), "s"((unsigned)c * 128u) : "memory");
  };

  // ---- per-lane LDS read offsets.  A swizzled piece index is (piece & ~15) | ((piece & 15) ^ key(row)), and piece = 2*step + h,
  // so base ^ (step << 5) walks the steps (the XORed bits never reach the row part of the address)
  const unsigned base1 = (unsigned)pair_w1_off(WD, px, h) + role * (32 * WD * 2);  // W2c stage: row 32*role + px
  const unsigned base2 = (unsigned)pair_w2_off(px, h) + role * (G2H * 32 * 128);   // W2a stage: rows of this role's output channels
  unsigned offs[4];  // shortcut buffer: own pixel row, 16-byte piece 4*role + j, this lane half's 8 bytes of it
  {
    const int rl_own = 32 * pg + px;
#pragma unroll
    for (int j = 0; j < 4; ++j) offs[j] = rl_own * 128 + (((4 * role + j) ^ ((rl_own >> 1) & 7)) * 16) + h * 8;
  }
  const unsigned xb_own = XB + ((pg * 2 + role) * 2) * 1024 + lane16, xb_other = XB + ((pg * 2 + (1 - role)) * 2) * 1024 + lane16;

  pf32x16 zacc[G2H];
#pragma unroll
  for (int g = 0; g < G2H; ++g)
#pragma unroll
    for (int i = 0; i < 16; ++i) zacc[g][i] = 0.f;
  pf32x16 sacc;
  unsigned xown[2][4], xnext[2][4];  // this role's two K steps of the chunk GEMM 2 consumes next / of the chunk after it

  pf16x8 dwf1[DF], dwf2[DF], dxop;
  auto flush1 = [&]() {  // the held-back tail of GEMM 1
#pragma unroll
    for (int i = 0; i < DF; ++i) sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(dwf1[i], breg[KS1 - DF + i], sacc, 0, 0, 0);
  };
  auto flush2 = [&]() {  // ... of GEMM 2 (the last K step: one MFMA per output fragment of this role)
#pragma unroll
    for (int i = 0; i < DF; ++i) zacc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dwf2[i], dxop, zacc[i], 0, 0, 0);
  };
  // GEMM 1 of this role's fragment of one chunk from ring slot `slot` -> sacc.  One step = one MFMA, pinned by a sched_barrier:
  // the fragment read LA steps ahead, the MFMA, and every few steps one piece of the period's memory work
  auto gemm1 = [&](int slot, auto&& pre, auto&& piece, auto np_tag) {
    constexpr int NM = KS1, NP = decltype(np_tag)::value, STRIDE = NP ? (NM - DF) / NP : NM;
    const unsigned char* st = lds + RING + slot * SB;
    pf16x8 wf[NM];
#pragma unroll
    for (int n = 0; n < LA; ++n) wf[n] = *reinterpret_cast<const pf16x8*>(st + (base1 ^ ((n & 7) << 5)) + (n >> 3) * 256);
    PAIR_FENCE();
    pre();  // the previous period's held-back MFMAs cover the latency of the reads above
    PAIR_FENCE();
#pragma unroll
    for (int n = 0; n < NM - DF; ++n) {
      if (n + LA < NM) {
        const int m = n + LA;
        if (PAIR_ABL & 16) wf[m] = wf[m - LA];
        else wf[m] = *reinterpret_cast<const pf16x8*>(st + (base1 ^ ((m & 7) << 5)) + (m >> 3) * 256);
      }
      if (n == 0) {
        pf32x16 zero;
#pragma unroll
        for (int i = 0; i < 16; ++i) zero[i] = 0.f;
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[n], breg[n], zero, 0, 0, 0);
      } else {
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[n], breg[n], sacc, 0, 0, 0);
      }
      if (!(PAIR_ABL & 1) && NP && n % STRIDE == 0 && n / STRIDE < NP) piece(n / STRIDE);
      PAIR_FENCE();
    }
#pragma unroll
    for (int i = 0; i < DF; ++i) dwf1[i] = wf[NM - DF + i];
  };
  // epilogue 1 of chunk c, this role's fragment, in parts j (4 channels of the lane's pixel each), split into the pieces the
  // pipeline places one by one
  pu32x2 e_s8[2];
  pf32x4 e_a4[2], e_b4[2];
  float e_v[4];
  auto epi_load = [&](int c, int j) {  // the part's shortcut values and constants: LDS -> registers (issued one part ahead)
    const unsigned char* sb = lds + SC + (c & 1) * 16384;
    const unsigned char* cb = lds + CONSTS + (c * 64 + 32 * role + 4 * h) * 4;
    e_s8[j & 1] = *reinterpret_cast<const pu32x2*>(sb + offs[j]);
    e_a4[j & 1] = *reinterpret_cast<const pf32x4*>(cb + 8 * j * 4);
    e_b4[j & 1] = *reinterpret_cast<const pf32x4*>(cb + NC * 4 + 8 * j * 4);
  };
  auto epi_calc = [&](int j, int half) {  // two of the part's four values: acc * a + b, + shortcut (conversion folded into the add)
    const unsigned sh = e_s8[j & 1][half];
    e_v[2 * half] = pair_add_half_lo(sh, __builtin_fmaf(sacc[4 * j + 2 * half], e_a4[j & 1][2 * half], e_b4[j & 1][2 * half]));
    e_v[2 * half + 1] = pair_add_half_hi(sh, __builtin_fmaf(sacc[4 * j + 2 * half + 1], e_a4[j & 1][2 * half + 1], e_b4[j & 1][2 * half + 1]));
  };
  auto epi_store = [&](int c, int j, unsigned (&xn)[2][4]) {  // round, ReLU, keep (GEMM 2's operand), put back into the shortcut buffer
    const pf16x2 lo = {(_Float16)e_v[0], (_Float16)e_v[1]}, hi = {(_Float16)e_v[2], (_Float16)e_v[3]};
    const unsigned ulo = pair_relu_pk(__builtin_bit_cast(unsigned, lo)), uhi = pair_relu_pk(__builtin_bit_cast(unsigned, hi));
    // accumulator register i = 4j + r  ->  K step (j >> 1) of this role's two, slots t = 4*(j&1) + r
    xn[j >> 1][2 * (j & 1)] = ulo;
    xn[j >> 1][2 * (j & 1) + 1] = uhi;
    *reinterpret_cast<pu32x2*>(lds + SC + (c & 1) * 16384 + offs[j]) = pu32x2{ulo, uhi};
  };
  // GEMM 2 of one chunk from ring slot `slot`: K steps [own 0, own 1, partner 0, partner 1] x this role's G2H output fragments;
  // EPI: epilogue 1 of chunk cn rides in the MFMAs' shadow -> xn
  auto gemm2 = [&](int slot, const unsigned (&xo)[2][4], auto&& pre, auto epi_tag, int cn, unsigned (&xn)[2][4], auto&& piece, auto np_tag) {
    constexpr bool EPI = decltype(epi_tag)::value && !(PAIR_ABL & 2);
    constexpr int NM = 4 * G2H;
    constexpr int NP = decltype(np_tag)::value, STRIDE = NP ? (NM - DF) / NP : NM;
    constexpr int SP = (NM - DF) / 4;  // steps per epilogue part (3 or 1: the period's own steps are NM - DF)
    const unsigned char* st = lds + RING + slot * SB;
    pf16x8 xop[4];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      xop[q] = __builtin_bit_cast(pf16x8, pu32x4{xown[q][0], xown[q][1], xown[q][2], xown[q][3]});
      xop[2 + q] = __builtin_bit_cast(pf16x8, pu32x4{xo[q][0], xo[q][1], xo[q][2], xo[q][3]});
    }
    // step n: K step index q = n / G2H in the order above = chunk step s4 = 2*role + q (own) or 2*(1-role) + (q-2) (partner)
    const unsigned s4x[4] = {(unsigned)(2 * role) << 5, (unsigned)(2 * role + 1) << 5, (unsigned)(2 * (1 - role)) << 5, (unsigned)(2 * (1 - role) + 1) << 5};
    pf16x8 wf[NM];
#pragma unroll
    for (int n = 0; n < LA; ++n) wf[n] = *reinterpret_cast<const pf16x8*>(st + (base2 ^ s4x[n / G2H]) + (n % G2H) * 32 * 128);
    if constexpr (EPI) epi_load(cn, 0);
    PAIR_FENCE();
    pre();  // GEMM 1's held-back tail: covers the latency of the reads above and completes sacc for the epilogue below
    PAIR_FENCE();
#pragma unroll
    for (int n = 0; n < NM - DF; ++n) {
      if (n + LA < NM) {
        const int m = n + LA;
        if (PAIR_ABL & 16) wf[m] = wf[m - LA];
        else wf[m] = *reinterpret_cast<const pf16x8*>(st + (base2 ^ s4x[m / G2H]) + (m % G2H) * 32 * 128);
      }
      zacc[n % G2H] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[n], xop[n / G2H], zacc[n % G2H], 0, 0, 0);
      if constexpr (EPI) {
        const int j = n / SP, q = n % SP;
        if (j < 4) {
          if (SP >= 3) {
            if (q == 0) {
              if (j + 1 < 4) epi_load(cn, j + 1);
              epi_calc(j, 0);
            }
            if (q == 1) epi_calc(j, 1);
            if (q == 2) epi_store(cn, j, xn);
          } else {  // one step per part
            if (j + 1 < 4) epi_load(cn, j + 1);
            epi_calc(j, 0);
            epi_calc(j, 1);
            epi_store(cn, j, xn);
          }
        }
      }
      if (!(PAIR_ABL & 1) && NP && n % STRIDE == 0 && n / STRIDE < NP) piece(n / STRIDE);
      PAIR_FENCE();
    }
#pragma unroll
    for (int i = 0; i < DF; ++i) dwf2[i] = wf[NM - DF + i];
    dxop = xop[3];
  };
  auto put_x = [&]() {  // this role's half of the chunk -> exchange buffer (behind a barrier: the partner is done with the previous one)
    *reinterpret_cast<pu32x4*>(lds + xb_own) = pu32x4{xown[0][0], xown[0][1], xown[0][2], xown[0][3]};
    *reinterpret_cast<pu32x4*>(lds + xb_own + 1024) = pu32x4{xown[1][0], xown[1][1], xown[1][2], xown[1][3]};
  };
  auto get_x = [&](unsigned (&xo)[2][4]) {
    const pu32x4 a0 = *reinterpret_cast<const pu32x4*>(lds + xb_other), a1 = *reinterpret_cast<const pu32x4*>(lds + xb_other + 1024);
#pragma unroll
    for (int e = 0; e < 4; ++e) xo[0][e] = a0[e], xo[1][e] = a1[e];
  };
  using TagT = std::true_type;
  using TagF = std::false_type;
  using NP0 = std::integral_constant<int, 0>;
  using NP1 = std::integral_constant<int, NWI>;
  using NP2 = std::integral_constant<int, NWI + 2>;

  // Order of this wave's vector-memory operations (vmcnt retires in order; every wait below leaves NVM = NWI + 2 in flight):
  //   W(0) SC(0) W(1) SC(1) | t=0: W(2) | t=1: Y(0)x2 W(3) | t=2: W(4) SC(2) | t=3: Y(1)x2 W(5) | t=4: W(6) SC(3) | ...
  // period t needs stage W(t) (and, t even, the shortcut chunk requested with W(t)): everything younger is NWI + 2 operations.
  stamp();
  dma_w(0, 0);
  dma_sc(0);
  dma_w(1, 1);
  dma_sc(1);
  // ================= t = 0: GEMM 1 and epilogue 1 of chunk 0
  PAIR_FENCE();
  pair_wait_vm<NVM>();
  pair_barrier();
  PAIR_FENCE();
  if (role) __builtin_amdgcn_s_setprio(1);  // the younger wave of each SIMD loses every arbitration at equal priority
  gemm1(0, [] {}, [&](int i) { dma_w_piece(2, 2, i); }, NP1{});
  PAIR_FENCE();
  flush1();
  PAIR_FENCE();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    epi_load(0, j);
    epi_calc(j, 0);
    epi_calc(j, 1);
    epi_store(0, j, xown);
  }
  PAIR_FENCE();
  int slot = 1;  // ring slot of stage t = 2c + 1
  for (int c = 0; c < NCH - 1; ++c) {
    const int slot1 = slot + 1 >= 3 ? slot - 2 : slot + 1;  // stage 2c+2
    const int slot2 = slot1 + 1 >= 3 ? slot1 - 2 : slot1 + 1;  // stage 2c+3 (= the slot stage 2c sat in)
    // ================= t = 2c+1: X(c), Y(c) leave; GEMM 1 of chunk c+1
    PAIR_FENCE();
    stamp();
    pair_wait_vm<NVM>();
    stamp();
    pair_barrier();
    stamp();
    PAIR_FENCE();
    if (!(PAIR_ABL & 8)) put_x();
    if (!(PAIR_ABL & 4)) store_y(c);
    PAIR_FENCE();
    if (c == 0) gemm1(slot, [] {}, [&](int i) { dma_w_piece(2 * c + 3, slot2, i); }, NP1{});
    else gemm1(slot, flush2, [&](int i) { dma_w_piece(2 * c + 3, slot2, i); }, NP1{});
    PAIR_FENCE();
    // ================= t = 2c+2: GEMM 2 of chunk c || epilogue 1 of chunk c+1
    stamp();
    pair_wait_vm<NVM>();
    stamp();
    pair_barrier();
    stamp();
    PAIR_FENCE();
    unsigned xo[2][4];
    if (!(PAIR_ABL & 8)) get_x(xo);
    else
      for (int a = 0; a < 2; ++a)
        for (int e = 0; e < 4; ++e) xo[a][e] = xown[a][e];
    PAIR_FENCE();
    gemm2(slot1, xo, flush1, TagT{}, c + 1, xnext, [&](int i) {
      if (i < NWI) dma_w_piece(2 * c + 4, slot, i);
      else dma_sc_piece(c + 2, i - NWI);
    }, NP2{});
    PAIR_FENCE();
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int e = 0; e < 4; ++e) xown[a][e] = xnext[a][e];
    slot = slot2;
  }
  // ================= t = 2*NCH - 1: X, Y of the last chunk leave
  PAIR_FENCE();
  pair_wait_vm<NVM>();
  pair_barrier();
  PAIR_FENCE();
  put_x();
  store_y(NCH - 1);
  // ================= t = 2*NCH: GEMM 2 of the last chunk
  PAIR_FENCE();
  pair_wait_vm<0>();  // stage 2*NCH - 1 has been in flight for a whole period; nothing else is requested any more
  pair_barrier();
  PAIR_FENCE();
  {
    unsigned xo[2][4];
    get_x(xo);
    PAIR_FENCE();
    gemm2(slot, xo, flush2, TagF{}, 0, xnext, [](int) {}, NP0{});
  }
  PAIR_FENCE();
  flush2();
  PAIR_FENCE();
  stamp();

  // ---- epilogue 2: Z = relu(Zacc * a2 + b2), float16: this role's WD/2 output channels
  {
    const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(pr.z, 0, (unsigned)M * WD * 2, 0x00020000);
    const unsigned z_vo = row < M ? (unsigned)row * (WD * 2) + (role * (WD / 2) + 4 * h) * 2 : kPairOOB;
    const unsigned char* cb = lds + CONSTS2 + (role * (WD / 2) + 4 * h) * 4;
#pragma unroll
    for (int g = 0; g < G2H; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const pf32x4 a4 = *reinterpret_cast<const pf32x4*>(cb + (32 * g + 8 * j) * 4);
        const pf32x4 b4 = *reinterpret_cast<const pf32x4*>(cb + WD * 4 + (32 * g + 8 * j) * 4);
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = zacc[g][4 * j + r] * a4[r] + b4[r];
          if (ka.relu2) v[r] = fmaxf(v[r], 0.f);
        }
        const pf16x2 lo = {(_Float16)v[0], (_Float16)v[1]}, hi = {(_Float16)v[2], (_Float16)v[3]};
        __builtin_amdgcn_raw_buffer_store_b64(pu32x2{__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)}, rz,
                                              z_vo + (32 * g + 8 * j) * 2, 0, 0);
      }
  }
}

// =====================================================================================================================
// The FIRST layer alone in the same persistent form (review item 2 of round 4: "a persistent form of the bandwidth-class layers"):
//     Y = relu( (B · W2cᵀ) ∘ a1 + b1 + S )
// A workgroup keeps its 128 pixels of B in registers and walks ALL 4·WD output channels in chunks of 64: per chunk one period —
// GEMM 1 of chunk c (this role's fragment: WD/16 MFMAs) beside the epilogue of chunk c-1 — with the filter chunk, the shortcut
// chunk (three buffers: requested two periods ahead) and the stores of Y (chunk c-2, whole 128-byte rows out of the shortcut buffer
// the epilogue wrote them back into) spread over the period's MFMA steps.  It reads the pair kernel's filter image (the W2c stages).
template <int WD>
__global__ __launch_bounds__(512, 1) void p2c_gemm_kernel(const PairArgs ka) {
  constexpr int NC = 4 * WD, NCH = NC / 64, SB = 128 * WD, NWI = SB / 8192, KS1 = WD / 16;
  constexpr int RING = 0, SC = 3 * SB, CONSTS = SC + 3 * 16384;
  constexpr int LA = 4;
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int pg = wave & 3, role = wave >> 2;
  const int px = lane & 31, h = lane >> 5;
  int k = 0;
  const int tile = blockIdx.x;
  for (int i = 1; i < kMaxPairProblems; ++i)
    if (i < ka.nprob && tile >= ka.prob[i].tile0) k = i;
  const PairProblem pr = ka.prob[k];
  const int m0 = (tile - pr.tile0) * 128;
  const int M = pr.M;
  const int row = m0 + pg * 32 + px;
  {
    const pf32x4* src = reinterpret_cast<const pf32x4*>(ka.ab1);
    pf32x4* dst = reinterpret_cast<pf32x4*>(lds + CONSTS);
    for (int i = threadIdx.x; i < 2 * NC / 4; i += 512) dst[i] = src[i];
  }
  pf16x8 breg[KS1];
  {
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(pr.b), 0, (unsigned)M * WD * 2, 0x00020000);
    const unsigned vo = (unsigned)row * (WD * 2) + h * 16;
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks)
      breg[ks] = __builtin_bit_cast(pf16x8, __builtin_amdgcn_raw_buffer_load_b128(rb, row < M ? vo + ks * 32 : kPairOOB, 0, 0));
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) asm volatile("" ::"v"(breg[ks]));
  }
  pair_wait_vm<0>();
  __syncthreads();

  const pi32x4 rw = pair_rsrc_words(ka.w, (unsigned)(2 * NCH) * SB);
  const pi32x4 rs = pair_rsrc_words(pr.s, (unsigned)M * NC * 2);
  const pi32x4 ryw = pair_rsrc_words(pr.y, (unsigned)M * NC * 2);
  const unsigned lane16 = lane * 16;
  unsigned sc_vo[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int rl = 32 * pg + 16 * role + 8 * i + (lane >> 3);
    const int piece = (lane & 7) ^ ((rl >> 1) & 7);
    sc_vo[i] = (m0 + rl < M) ? (unsigned)(m0 + rl) * (NC * 2) + piece * 16 : kPairOOB;
  }
  const unsigned sc_row0 = (32 * pg + 16 * role) * 128;
  // chunk c's filter stage inside the pair image (its order of consumption: W2c[0], then (W2c[c+1], W2a[c]) ...)
  auto dma_w_piece = [&](int c, int slot, int i) {
    const int stage = c == 0 ? 0 : 2 * c - 1;
    const unsigned piece = (unsigned)(wave + 8 * i) * 1024u;
    pair_dma16(rw, RING + slot * SB + piece, c < NCH ? lane16 : kPairOOB, (unsigned)stage * SB + piece);
  };
  auto dma_sc_piece = [&](int c, int buf, int i) {
    pair_dma16(rs, SC + buf * 16384 + sc_row0 + i * 1024, c < NCH ? sc_vo[i] : kPairOOB, (unsigned)c * 128u);
  };
  auto store_y = [&](int c, int buf) {
    const unsigned char* sb = lds + SC + buf * 16384 + sc_row0 + lane16;
    const pu32x4 v0 = *reinterpret_cast<const pu32x4*>(sb), v1 = *reinterpret_cast<const pu32x4*>(sb + 1024);
    asm volatile("s_nop 0\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen" ::"v"(v0), "v"(sc_vo[0]), "s"(ryw), "s"((unsigned)c * 128u) : "memory");
    asm volatile("s_nop 0\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen" ::"v"(v1), "v"(sc_vo[1]), "s"(ryw), "s"((unsigned)c * 128u) : "memory");
  };
  const unsigned base1 = (unsigned)pair_w1_off(WD, px, h) + role * (32 * WD * 2);
  unsigned offs[4];
  {
    const int rl_own = 32 * pg + px;
#pragma unroll
    for (int j = 0; j < 4; ++j) offs[j] = rl_own * 128 + (((4 * role + j) ^ ((rl_own >> 1) & 7)) * 16) + h * 8;
  }
  pf32x16 sacc[2];
  pu32x2 e_s8[2];
  pf32x4 e_a4[2], e_b4[2];
  float e_v[4];
  if (role) __builtin_amdgcn_s_setprio(1);

  // order of this wave's vector-memory operations: W(0) | SC(0) W(1) | c=0: SC(1) W(2) | c=1: SC(2) W(3) | c=2: Y(0)x2 SC(3) W(4) | ...
  // period c needs W(c) and SC(c-1), both requested in period c-2: everything younger is what period c-1 issued (6 operations, 8 from c = 3 on)
#pragma unroll
  for (int i = 0; i < NWI; ++i) dma_w_piece(0, 0, i);
  dma_sc_piece(0, 0, 0);
  dma_sc_piece(0, 0, 1);
#pragma unroll
  for (int i = 0; i < NWI; ++i) dma_w_piece(1, 1, i);
  int slot = 0, buf = 0;  // ring slot of chunk c's filters; shortcut buffer of chunk c
  // one period (compile-time shape: with / without GEMM 1, with / without the epilogue of the chunk before)
  auto run = [&](int c, auto mm_tag, auto epi_tag, pf32x16& acc_mm, const pf32x16& acc_epi) {
    constexpr bool MM = decltype(mm_tag)::value, EPI = decltype(epi_tag)::value;
    const int slot2 = slot + 2 >= 3 ? slot - 1 : slot + 2;   // chunk c+2's filters: the slot chunk c-1 sat in
    const int buf1 = buf + 1 >= 3 ? buf - 2 : buf + 1;       // chunk c+1's shortcut: the buffer chunk c-2 sat in
    const int bufm1 = buf - 1 < 0 ? buf + 2 : buf - 1;       // chunk c-1's
    PAIR_FENCE();
    if (c < 3) pair_wait_vm<NWI + 2>();
    else pair_wait_vm<NWI + 4>();
    pair_barrier();
    PAIR_FENCE();
    if (c >= 2) store_y(c - 2, buf1);  // finished in place by both roles during period c-1
    PAIR_FENCE();
    const unsigned char* st = lds + RING + slot * SB;
    const unsigned char* sb = lds + SC + bufm1 * 16384;
    const unsigned char* cb = lds + CONSTS + ((c - 1) * 64 + 32 * role + 4 * h) * 4;
    pf16x8 wf[KS1];
    if constexpr (MM) {
#pragma unroll
      for (int n = 0; n < LA; ++n) wf[n] = *reinterpret_cast<const pf16x8*>(st + (base1 ^ ((n & 7) << 5)) + (n >> 3) * 256);
    }
    auto epi_load = [&](int j) {
      e_s8[j & 1] = *reinterpret_cast<const pu32x2*>(sb + offs[j]);
      e_a4[j & 1] = *reinterpret_cast<const pf32x4*>(cb + 8 * j * 4);
      e_b4[j & 1] = *reinterpret_cast<const pf32x4*>(cb + NC * 4 + 8 * j * 4);
    };
    auto epi_calc = [&](int j, int half) {
      const unsigned sh = e_s8[j & 1][half];
      e_v[2 * half] = pair_add_half_lo(sh, __builtin_fmaf(acc_epi[4 * j + 2 * half], e_a4[j & 1][2 * half], e_b4[j & 1][2 * half]));
      e_v[2 * half + 1] = pair_add_half_hi(sh, __builtin_fmaf(acc_epi[4 * j + 2 * half + 1], e_a4[j & 1][2 * half + 1], e_b4[j & 1][2 * half + 1]));
    };
    auto epi_store = [&](int j) {
      const pf16x2 lo = {(_Float16)e_v[0], (_Float16)e_v[1]}, hi = {(_Float16)e_v[2], (_Float16)e_v[3]};
      *reinterpret_cast<pu32x2*>(lds + SC + bufm1 * 16384 + offs[j]) =
          pu32x2{pair_relu_pk(__builtin_bit_cast(unsigned, lo)), pair_relu_pk(__builtin_bit_cast(unsigned, hi))};
    };
    if constexpr (EPI) epi_load(0);
    PAIR_FENCE();
    constexpr int SP = KS1 / 4;  // steps per epilogue part (4 or 2)
#pragma unroll
    for (int n = 0; n < KS1; ++n) {
      if constexpr (MM) {
        if (n + LA < KS1) {
          const int m = n + LA;
          wf[m] = *reinterpret_cast<const pf16x8*>(st + (base1 ^ ((m & 7) << 5)) + (m >> 3) * 256);
        }
        if (n == 0) {
          pf32x16 zero;
#pragma unroll
          for (int i = 0; i < 16; ++i) zero[i] = 0.f;
          acc_mm = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[n], breg[n], zero, 0, 0, 0);
        } else {
          acc_mm = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[n], breg[n], acc_mm, 0, 0, 0);
        }
      }
      if constexpr (EPI) {
        const int j = n / SP, q = n % SP;
        if (SP == 4) {
          if (q == 0 && j + 1 < 4) epi_load(j + 1);
          if (q == 1) epi_calc(j, 0);
          if (q == 2) epi_calc(j, 1);
          if (q == 3) epi_store(j);
        } else {
          if (q == 0) {
            if (j + 1 < 4) epi_load(j + 1);
            epi_calc(j, 0);
          } else {
            epi_calc(j, 1);
            epi_store(j);
          }
        }
      }
      // requests: this wave's 2 shortcut pieces of chunk c+1 and NWI filter pieces of chunk c+2, one every other step
      if ((n & 1) == 0 && n / 2 < NWI + 2) {
        const int i = n / 2;
        if (i < 2) dma_sc_piece(c + 1, buf1, i);
        else dma_w_piece(c + 2, slot2, i - 2);
      }
      PAIR_FENCE();
    }
    slot = slot + 1 >= 3 ? slot - 2 : slot + 1;
    buf = buf1;
  };
  using TT = std::true_type;
  using FF = std::false_type;
  run(0, TT{}, FF{}, sacc[0], sacc[1]);
  for (int c = 1; c + 1 < NCH; c += 2) {
    run(c, TT{}, TT{}, sacc[1], sacc[0]);
    run(c + 1, TT{}, TT{}, sacc[0], sacc[1]);
  }
  run(NCH - 1, TT{}, TT{}, sacc[1], sacc[0]);
  run(NCH, FF{}, TT{}, sacc[0], sacc[1]);
  run(NCH + 1, FF{}, FF{}, sacc[0], sacc[1]);
  pair_wait_vm<0>();
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
bool pair_supported(int WD) { return WD == 128 || WD == 256; }
size_t pair_packed_halves(int WD) { return (size_t)(4 * WD / 64) * 2 * pair_stage_bytes(WD) / 2; }
size_t pair_lds_bytes(int WD) { return 3 * (size_t)pair_stage_bytes(WD) + 2 * 16384 + 16384 + (size_t)4 * WD * 8 + (size_t)WD * 8; }

static unsigned short pair_f2h(float f) {
  const _Float16 hv = (_Float16)f;  // round to nearest even, as v_cvt_f16_f32
  unsigned short u;
  std::memcpy(&u, &hv, 2);
  return u;
}

// w1: [4*WD][WD] (branch2c, K contiguous), w2: [WD][4*WD] (next branch2a), float (already carrying the per-row power-of-two
// scaling of the float16 path) -> the stage images the kernel streams, as float16 bit patterns
void pair_pack_filters(const float* w1, const float* w2, int WD, unsigned short* out) {
  const int NC = 4 * WD, NCH = NC / 64, SB = pair_stage_bytes(WD);
  // stage order = the kernel's order of consumption: W2c[0], then (W2c[c+1], W2a[c]) for c = 0 .. NCH-2, then W2a[NCH-1]
  for (int c = 0; c < NCH; ++c) {
    const int t1 = c == 0 ? 0 : 2 * c - 1, t2 = c == NCH - 1 ? 2 * NCH - 1 : 2 * c + 2;
    unsigned short* s1 = out + (size_t)t1 * SB / 2;
    for (int r = 0; r < 64; ++r)
      for (int q = 0; q < WD / 8; ++q) {
        unsigned short* d = s1 + pair_w1_off(WD, r, q) / 2;
        for (int e = 0; e < 8; ++e) d[e] = pair_f2h(w1[(size_t)(64 * c + r) * WD + 8 * q + e]);
      }
    unsigned short* s2 = out + (size_t)t2 * SB / 2;
    for (int r = 0; r < WD; ++r)
      for (int q = 0; q < 8; ++q) {
        unsigned short* d = s2 + pair_w2_off(r, q) / 2;
        for (int t = 0; t < 8; ++t) d[t] = pair_f2h(w2[(size_t)r * NC + 64 * c + pair_chan(q >> 1, q & 1, t)]);
      }
  }
}

long pair_grid(PairArgs& a) {  // fills prob[].tile0; returns the number of workgroups
  long t = 0;
  for (int i = 0; i < a.nprob; ++i) {
    a.prob[i].tile0 = (int)t;
    t += (a.prob[i].M + 127) / 128;
  }
  return t;
}

int launch_pair_gemm(const PairArgs& a, long grid, void* stream) {
  if (!pair_supported(a.WD) || a.nprob < 1 || a.nprob > kMaxPairProblems || grid <= 0) return (int)hipErrorInvalidValue;
  for (int i = 0; i < a.nprob; ++i)
    if ((long)a.prob[i].M * 4 * a.WD * 2 >= (1L << 31)) return (int)hipErrorInvalidValue;  // 32-bit buffer offsets
  const size_t ldsb = pair_lds_bytes(a.WD);
  static bool attr_set[2] = {false, false};
  hipError_t e = hipSuccess;
  if (a.WD == 256) {
    if (!attr_set[0]) {
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pair_gemm_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
      if (e != hipSuccess) return (int)e;
      attr_set[0] = true;
    }
    if (a.dbg) {
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pair_gemm_kernel<256, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
      if (e != hipSuccess) return (int)e;
      hipLaunchKernelGGL((pair_gemm_kernel<256, true>), dim3((unsigned)grid), dim3(512), ldsb, (hipStream_t)stream, a);
    } else {
      hipLaunchKernelGGL(pair_gemm_kernel<256>, dim3((unsigned)grid), dim3(512), ldsb, (hipStream_t)stream, a);
    }
  } else {
    if (!attr_set[1]) {
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pair_gemm_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
      if (e != hipSuccess) return (int)e;
      attr_set[1] = true;
    }
    if (a.dbg) {
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pair_gemm_kernel<128, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
      if (e != hipSuccess) return (int)e;
      hipLaunchKernelGGL((pair_gemm_kernel<128, true>), dim3((unsigned)grid), dim3(512), ldsb, (hipStream_t)stream, a);
    } else {
      hipLaunchKernelGGL(pair_gemm_kernel<128>, dim3((unsigned)grid), dim3(512), ldsb, (hipStream_t)stream, a);
    }
  }
  return (int)hipGetLastError();
}

size_t p2c_lds_bytes(int WD) { return 3 * (size_t)pair_stage_bytes(WD) + 3 * 16384 + (size_t)4 * WD * 8; }
int launch_p2c_gemm(const PairArgs& a, long grid, void* stream) {
  if (!pair_supported(a.WD) || a.nprob < 1 || a.nprob > kMaxPairProblems || grid <= 0) return (int)hipErrorInvalidValue;
  const size_t ldsb = p2c_lds_bytes(a.WD);
  hipError_t e;
  if (a.WD == 256) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&p2c_gemm_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(p2c_gemm_kernel<256>, dim3((unsigned)grid), dim3(512), ldsb, (hipStream_t)stream, a);
  } else {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&p2c_gemm_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(p2c_gemm_kernel<128>, dim3((unsigned)grid), dim3(512), ldsb, (hipStream_t)stream, a);
  }
  return (int)hipGetLastError();
}

}  // namespace dc
