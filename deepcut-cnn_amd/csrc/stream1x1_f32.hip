// stream1x1_f32.hip — the float32 form of stream1x1.hip ("ws1x1f"): the dense 1x1 layers with 64 / 128 / 256 / 512 input channels — above all
// the expansions of a bottleneck block (+ shortcut + ReLU, resNx_branch2c of ResNet-152.prototxt; reference: one SGEMM per image,
// base_conv_layer.cpp:326-341) — with the filters resident in registers and the pixels walked in 16-pixel steps.
//
// Why, in float32, where these layers are MFMA-bound and not byte-bound: at batch 1 (BASELINE configs[1], the headline) a conv4_x layer has
// M = 34 x 46 = 1 564 pixels = 4 x 17 x 23, and every tiling of it into 32-row MFMA fragments leaves 56-60 of the 256 CUs without a
// workgroup (196 or 200 workgroups: DESIGN.md 4.1, EXPERIMENTS.md A "why the one-forward figure sits where it does"); inside a workgroup
// 4-7 k of ~25 k cycles are prologue / split-K exchange / epilogue.  Here the unit of work is a 16-pixel x 64-channel step of
// v_mfma_f32_16x16x4_f32 (the same 64 flops per cycle and SIMD as the 32x32x2 form), the grid is ONE workgroup per CU — 16 channel
// slices x 16 pixel ranges at N = 1024 — and a workgroup walks 6 or 7 of the 98 steps of its range: 7 / 6.125 = 87.5 % of the chip's matrix
// cycles are used where 196 / 256 = 76.6 % were, the filters (64 KB per workgroup) are fetched once, and there is no split-K exchange.
//
// Per workgroup (4 waves = 4 x 16 channels; a wave alone on its SIMD at K >= 256, two workgroups per CU at K <= 128):
//  * filters: registers, ROW operand (rows = 16 channels), K/4 of them per lane.  Lane (c, q) of the 16x16x4 instruction supplies
//    A[c][q] and B[q][p]: with the K range cut into four runs of K/4 — lane q takes run q, matrix step m is element m of every run — one
//    ds_read_b128 of the pixel's row feeds four matrix steps and the filters load as 16-byte vectors (stream1x1f_pack_filters);
//  * pixels: LDS-DMA ring of D stages of 16 rows in 1 KiB requests (a pixel's row at K = 256; several rows at K = 128 / 64: see the
//    kernel), padded by 16 bytes per row or row block: conflict-free reads whose 16 offsets per step are immediates; counted vmcnt;
//  * ONE stream of matrix instructions from the first step's first product to the last step's last: a step's barrier (stage k+1 published,
//    the slot of step k-1 free), the ring's requests and the PREVIOUS step's epilogue sit between its products, the operand reads run
//    three reads ahead across the step boundary;
//  * two accumulators taken in turn (a matrix step does not wait for the one before it), added at the end;
//  * shortcut and output tiles (16 pixels x 16 channels = 1 KiB) through a wave-private LDS buffer as whole 64-byte runs, the epilogue's
//    affine / add / ReLU in fp32 in between.
// Not bit-identical to the gather-GEMM tiles (another summation grouping); same 1e-3 bound against the oracle, measured ~3e-6.
// Forms that were built and dropped (eight waves with four moving the bytes, the two waves of a SIMD sharing the products, the early-start
// prologue) and the timing ablations: profiles/r06_stream1x1f_forms.txt, EXPERIMENTS.md L; measurements: DESIGN.md 4.1h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "kernels.h"

namespace dc {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned kOOBf = 0x80000000u;
#pragma clang diagnostic ignored "-Winline-asm"

struct WsfArgs {
  const void* x;
  const void* w;  // stream1x1f_pack_filters image
  const float* scale;
  const float* shift;
  void* y;
  const void* resid;
  int M, Cout, sxb, ypb;  // pixels, channels, bytes between pixels of x / of y (and resid)
  int tn, J, S, sbase, srem;  // J pixel ranges of sbase steps, the first srem of them one more (those are dispatched first)
  unsigned div_tn[2];
  long long* dbg;
  // the stem form ("ws7x7f"): output map and image geometry (a 1 KiB request gathers an output pixel's 7 rows x 8 pixels of 4 channels)
  int OW, OHW, H, W, xrsb, ximgb;  // output width, output pixels per image, image rows / pixels per row, bytes between image rows / images
  unsigned div_ow[2], div_ohw[2];
};

__device__ __forceinline__ unsigned f_uni(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ i32x4 f_rsrc_words(const void* p) {
  const unsigned long long a = (unsigned long long)p;
  return i32x4{__builtin_amdgcn_readfirstlane((int)(unsigned)a), __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu)), 0x7fffffff, 0x00020000};
}
__device__ __forceinline__ void f_dma16(i32x4 rs, unsigned lds_, unsigned voff) {
  const unsigned lds = f_uni(lds_);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds), "v"(voff), "s"(rs) : "memory", "m0");
}
__device__ __forceinline__ void f_store16_untracked(i32x4 rs, unsigned voff, u32x4 v) {
  asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(rs) : "memory");
}
template <int N>
__device__ __forceinline__ void f_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void f_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ int f_fastdiv(int n, const unsigned (&mg)[2]) {
  return (mg[1] >> 31) ? n : (int)(__umulhi((unsigned)n, mg[0]) >> (mg[1] & 31));
}

// Counted waits.  The vector-memory requests of a wave, in issue order (vmcnt counts them all, stores included):
//   prologue: stage 0 .. D-2 (NA each), shortcut tile 0, shortcut tile 1 (NR each); [wait: stage 0 landed] barrier
//   step k, between its matrix products:
//             [wait: stage k+1 landed] barrier, stage k+D-1 (NA)
//             from step 1 on: [wait: shortcut tile k-1 landed] output tile k-1 (NS), shortcut tile k+1 (NR)
// A wait names how many requests issued AFTER the awaited one may still be in flight: the totals below, evaluated at compile time for the
// first steps (while the pipeline fills) and for the steady state.
template <int D, int NA, int NR, int NS>
struct ReqCount {
  static constexpr int kSteady = 8, kPeel = 3;
  static constexpr int pos(int k) { return k > 0 ? k : 0; }
  static constexpr int before(int k) { return (D - 1) * NA + 2 * NR + k * NA + pos(k - 1) * (NS + NR); }  // issued before step k
  static constexpr int end_stage(int j) { return j <= D - 2 ? (j + 1) * NA : before(j - D + 1) + NA; }
  static constexpr int end_resid(int r) { return r <= 1 ? (D - 1) * NA + (r + 1) * NR : before(r - 1) + NA + NS + NR; }
  static constexpr int pro = before(0) - end_stage(0);
  static constexpr int stage(int k) { return before(k) - end_stage(k + 1); }       // step k's wait for stage k + 1
  static constexpr int resid(int k) { return before(k) + NA - end_resid(k - 1); }  // step k's wait for shortcut tile k - 1
  static_assert(D - 1 <= kPeel, "the peeled steps cover the filling pipeline");
  static_assert(stage(kSteady) == (D - 3) * NA + (D - 2) * (NR + NS) && stage(kSteady) == stage(kPeel) && stage(kSteady) == stage(kSteady + 1), "steady from step D-1 on");
  static_assert(resid(kSteady) == 2 * NA + NS + NR && resid(kSteady) == resid(kPeel) && resid(kSteady) == resid(kSteady + 1), "steady from step 3 on");
  static_assert(resid(kSteady) <= 63 && stage(kSteady) >= 0 && pro >= 0, "vmcnt is a 6-bit counter");
};

// K = input channels (64, 128, 256 or 512), D = ring stages.  A stage is 16 pixel rows as 1 KiB requests: PPR requests per row at K >= 256,
// RPR = 2 / 4 rows per request at K = 128 / 64.  A request's 1 KiB is contiguous in LDS, so the 16 bytes of padding follow a BLOCK of RPR rows;
// inside a block row r keeps its 16-byte chunks rotated by r runs (chunk c of row r at position (c + r K/16) mod (K/4): the request's lane l
// simply fetches the chunk that belongs at position l), which puts the 16 lanes of a read — 16 rows, one chunk index — into 16 different
// bank groups again and leaves the read offsets immediates (a lane's run starts at ((q + r) mod 4) K bytes of its row instead of q K).
template <int K, int D, bool RES, bool RELU, bool STEM = false>
__global__ __launch_bounds__(256, 1) void ws1x1f_kernel(const WsfArgs a) {
  static_assert(!STEM || (K == 224 && !RES), "the stem form: 7 taps x 32 image floats per output pixel (the request's eighth tap is never read), no shortcut");
  const long long t_entry = (long long)__builtin_amdgcn_s_memrealtime();
  constexpr int RPR = K * 4 >= 1024 ? 1 : 1024 / (K * 4);  // pixel rows per 1 KiB request
  constexpr int PPR = K * 4 >= 1024 ? K * 4 / 1024 : 1;    // 1 KiB requests per pixel row
  constexpr int BLKB = (K * 4 >= 1024 ? K * 4 : 1024) + 16;  // bytes between padded blocks in LDS: a row (K >= 256: lanes p = 0..15 of a read land in 16 different bank groups) or RPR rows
  constexpr int STG = 16 / RPR * BLKB;  // a stage: 16 pixels
  constexpr int NA = 16 * PPR / RPR / 4;  // requests per wave and stage
  constexpr int NJ = K / 16;         // 16-byte reads (= 4 matrix steps each) per wave and step
  constexpr int NR = RES ? 1 : 0, NS = 1;
  static_assert((K == 64 || K == 128 || K % 256 == 0 || (STEM && K == 224)) && D >= 3 && D <= 4 && NA >= 1, "rows and requests divide each other; the ring");
  using Rq = ReqCount<D, NA, NR, NS>;
  constexpr int LDSB = D * STG + 4 * 2 * 1024;
  static_assert(LDSB <= 160 * 1024, "LDS of a CU");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[LDSB];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int p16 = lane & 15, q = lane >> 4;

  const int bx = blockIdx.x, xcd = bx & 7, bi = bx >> 3;
  const int ci = f_fastdiv(bi, a.div_tn);
  const int nt = bi - ci * a.tn;
  const int jc = xcd + 8 * ci;
  if (jc >= a.J) return;
  int gs0 = jc * a.sbase + min(jc, a.srem), gs1 = gs0 + a.sbase + (jc < a.srem ? 1 : 0);
  if (a.srem < 0) gs0 = a.S * jc / a.J, gs1 = a.S * (jc + 1) / a.J;  // (DC_WSF_REMAP=0: the longer ranges spread over the grid, for A/B timing)
  const int nw0 = nt * 64 + wave * 16;  // first channel of this wave

  // ---- epilogue constants (4 channels per lane: 4 q + r) and filters: untracked requests, the oldest of the wave (stream1x1.hip)
  f32x4 csc = {1.f, 1.f, 1.f, 1.f}, csh = {0.f, 0.f, 0.f, 0.f};
  {
    const unsigned co = (unsigned)((nw0 + 4 * q) * 4);
    if (a.scale) asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(csc) : "v"(co), "s"(f_rsrc_words(a.scale)) : "memory");
    if (a.shift) asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(csh) : "v"(co), "s"(f_rsrc_words(a.shift)) : "memory");
  }
  f32x4 wreg[NJ];
  {
    const i32x4 wrs = f_rsrc_words(a.w);
    const unsigned wl = (unsigned)lane * 16u;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const unsigned so = f_uni((unsigned)((((nw0 >> 4) * NJ) + j) * 1024));
      asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(wreg[j]) : "v"(wl), "s"(wrs), "s"(so) : "memory");
    }
  }

  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const i32x4 xr = f_rsrc_words(a.x), rr = f_rsrc_words(RES ? a.resid : a.y), yr = f_rsrc_words(a.y);
  const unsigned frag0 = (unsigned)(p16 / RPR * BLKB + p16 % RPR * K * 4 + ((q + p16 % RPR) & 3) * K);  // this lane's run of the pixel's row: + 16 j
  const unsigned obuf0 = (unsigned)(D * STG + wave * 2 * 1024);           // this wave's two 1 KiB tiles
  const unsigned ovec = (unsigned)((4 * p16 + q) * 16);                   // MFMA view: pixel p16, channels 4 q .. + 3
  const int orow = lane >> 2;                                             // memory view: pixel lane / 4, channels 4 (lane % 4) .. + 3
  const unsigned ooff = (unsigned)((nw0 + 4 * (lane & 3)) * 4);
  auto dma_req = [&](int gs, int slot, int i) {  // request i of the wave's NA for stage gs (issued whatever gs is: beyond the range it moves nothing)
    const int row0 = gs * 16;
    const int lim = gs < gs1 ? a.M - row0 : 0;
    const int pc = wave + 4 * i;  // request pc of the stage
    if (STEM) {  // output pixel row0 + pc: lane 8 tap + px fetches pixel (2 ox - 3 + px) of image row (2 oy - 3 + tap); tap 7 and the pad are zeros
      const int r = row0 + pc;
      const int n = f_fastdiv(r, a.div_ohw), rem = r - n * a.OHW;
      const int oy = f_fastdiv(rem, a.div_ow), ox = rem - oy * a.OW;
      const int tap = lane >> 3, iy = 2 * oy - 3 + tap, ix = 2 * ox - 3 + (lane & 7);
      const bool ok = pc < lim && tap < 7 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      const unsigned vo = ok ? (unsigned)n * (unsigned)a.ximgb + (unsigned)iy * (unsigned)a.xrsb + (unsigned)ix * 16u : kOOBf;
      f_dma16(xr, lds0 + (unsigned)(slot * STG + pc * BLKB), vo);
    } else if (RPR == 1) {        // part `part` of row `row`
      const int row = pc / PPR, part = pc - row * PPR;
      const unsigned vo = row < lim ? (unsigned)(row0 + row) * (unsigned)a.sxb + (unsigned)(part * 1024 + lane * 16) : kOOBf;
      f_dma16(xr, lds0 + (unsigned)(slot * STG + row * BLKB + part * 1024), vo);
    } else {                      // rows pc RPR .. + RPR - 1, K / 4 lanes each
      constexpr int LPR = 64 / RPR;  // lanes = 16-byte chunks per row
      const int r = lane / LPR, row = pc * RPR + r;
      const unsigned vo = row < lim ? (unsigned)(row0 + row) * (unsigned)a.sxb + (unsigned)(((lane - r * (LPR / 4)) & (LPR - 1)) * 16) : kOOBf;
      f_dma16(xr, lds0 + (unsigned)(slot * STG + pc * BLKB), vo);
    }
  };
  auto resid_step = [&](int gs, int buf) {  // NR requests, always
    if (!RES) return;
    const int row0 = gs * 16;
    const int lim = gs < gs1 ? a.M - row0 : 0;
    f_dma16(rr, lds0 + obuf0 + (unsigned)(buf * 1024), orow < lim ? (unsigned)(row0 + orow) * (unsigned)a.ypb + ooff : kOOBf);
  };
  auto stamp = [&](int sl) {
    if (a.dbg && lane == 0) {
      long long* d = a.dbg + ((long)blockIdx.x * 4 + wave) * 12;
      d[sl] = (long long)__builtin_readcyclecounter();
      if (sl == 0) d[8] = t_entry, d[10] = (long long)__builtin_amdgcn_s_memrealtime();
      if (sl == 7) d[9] = (long long)__builtin_amdgcn_s_memrealtime();
    }
  };
  stamp(0);
#pragma unroll
  for (int d = 0; d < D - 1; ++d)
#pragma unroll
    for (int i = 0; i < NA; ++i) dma_req(gs0 + d, d, i);
  resid_step(gs0, 0);
  resid_step(gs0 + 1, 1);
  stamp(1);
  f_wait_vm<Rq::pro>();
  asm volatile("" : "+v"(csc), "+v"(csh));
#pragma unroll
  for (int j = 0; j < NJ; ++j) asm volatile("" : "+v"(wreg[j]));
  f_lds_barrier();
  stamp(2);

  // One step = the NJ x 4 matrix products of 16 pixels, and everything else BETWEEN them, so that the matrix pipe runs from the first
  // step's first product to the last step's last one:
  //  * the barrier that publishes stage k+1 (and frees the slot of step k-1 for stage k+D-1, requested right behind it) sits inside step k;
  //    the operand reads run PD-1 reads ahead across the step boundary;
  //  * the epilogue of step k-1 (shortcut wait, LDS round trips, store, next shortcut request) rides in step k.
  constexpr int PD = NJ % 4 == 0 ? 4 : 7;  // (the stem's 14 reads per step: a ring of 7)
  static_assert(NJ % PD == 0, "the operand ring keeps its phase across steps");
  int slot = 0, ob = 0;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  f32x4 xf[PD];
#pragma unroll
  for (int j = 0; j < PD - 1; ++j) xf[j] = *reinterpret_cast<const f32x4*>(smem + frag0 + j * 16);
  auto epi_read = [&](unsigned char* obp, f32x4& rv) {
    if (RES) rv = *reinterpret_cast<const f32x4*>(obp + ovec);
  };
  auto epi_math = [&](unsigned char* obp, const f32x4& sum, const f32x4& rv) {  // register r of a lane = channel 4 q + r of pixel p16
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = sum[r] * csc[r] + csh[r];
      if (RES) v += rv[r];
      o[r] = RELU ? fmaxf(v, 0.f) : v;
    }
    *reinterpret_cast<f32x4*>(obp + ovec) = o;
  };
  auto epi_store = [&](int gs, const u32x4& ov) {  // memory view: 16 pixels x 64-byte runs
    const int row0 = gs * 16;
    f_store16_untracked(yr, orow < a.M - row0 ? (unsigned)(row0 + orow) * (unsigned)a.ypb + ooff : kOOBf, ov);
  };
  auto do_step = [&](int gs, auto k_tag) {
    constexpr int KS = decltype(k_tag)::value;  // the step's number while the request counts still change (Rq), -1 in the steady state
    constexpr int KQ = KS < 0 ? Rq::kSteady : KS;
    constexpr bool EPI = KS != 0;               // step 0 has no step before it
    // where the pieces sit among the step's NJ product groups (K = 64 has four: barrier behind the first, request + shortcut read in the
    // second, epilogue arithmetic + tile turn in the third, store + next shortcut request in the fourth)
    constexpr int SP = NJ >= 16 ? 2 : 1;
    constexpr int HB = NJ == 4 ? 0 : 1, H1 = NJ == 4 ? HB + NA : HB + NA + 1, H2 = H1 + SP, H3 = NJ == 4 ? H2 : H2 + SP, H4 = H3 + SP;
    static_assert(H4 < NJ && HB < NJ - PD + 1, "the pieces inside the product loop; stage k+1 is read only behind its barrier");
    f32x4 sum = {0.f, 0.f, 0.f, 0.f}, rv = {0.f, 0.f, 0.f, 0.f};
    u32x4 ov = {0u, 0u, 0u, 0u};
    if (EPI) sum = acc0 + acc1;
    unsigned char* const obp = smem + obuf0 + ob * 1024;
    const int nslot = slot + 1 == D ? 0 : slot + 1, pslot = slot == 0 ? D - 1 : slot - 1;
    const unsigned char* st = smem + slot * STG + frag0;
    const unsigned char* stn = smem + nslot * STG + frag0;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int jn = j + PD - 1;
      xf[jn % PD] = *reinterpret_cast<const f32x4*>(jn < NJ ? st + jn * 16 : stn + (jn - NJ) * 16);
      if (j == 0) {  // fresh accumulators: the first two products take the constant 0
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[j][0], xf[j % PD][0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[j][1], xf[j % PD][1], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      } else {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[j][0], xf[j % PD][0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[j][1], xf[j % PD][1], acc1, 0, 0, 0);
      }
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[j][2], xf[j % PD][2], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[j][3], xf[j % PD][3], acc1, 0, 0, 0);
      if (j == HB) {
        f_wait_vm<Rq::stage(KQ)>();
        asm volatile("s_barrier" ::: "memory");
      }
      if (j > HB && j <= HB + NA) dma_req(gs + D - 1, pslot, j - HB - 1);
      if (EPI && j == H1) {
        if (RES) f_wait_vm<Rq::resid(KQ)>();
        epi_read(obp, rv);
      }
      if (EPI && j == H2) epi_math(obp, sum, rv);
      if (EPI && j == H3) ov = *reinterpret_cast<const u32x4*>(obp + lane * 16);
      if (EPI && j == H4) {
        epi_store(gs - 1, ov);
        resid_step(gs + 1, ob);
        if (RES) ob ^= 1;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    slot = nslot;
  };
  int gs = gs0;
  if (gs < gs1) do_step(gs++, std::integral_constant<int, 0>{});
  if (gs < gs1) do_step(gs++, std::integral_constant<int, 1>{});
  if (gs < gs1) do_step(gs++, std::integral_constant<int, 2>{});
  static_assert(Rq::kPeel == 3, "steps 0, 1, 2 above");
  stamp(3);
  for (; gs < gs1; ++gs) do_step(gs, std::integral_constant<int, -1>{});
  stamp(4);
  // the last step's epilogue, alone
  f_wait_vm<0>();
  stamp(5);
  if (gs1 > gs0) {
    unsigned char* const obp = smem + obuf0 + ob * 1024;
    f32x4 rv = {0.f, 0.f, 0.f, 0.f};
    epi_read(obp, rv);
    epi_math(obp, acc0 + acc1, rv);
    epi_store(gs1 - 1, *reinterpret_cast<const u32x4*>(obp + lane * 16));
  }
  stamp(6);
  stamp(7);
}

typedef void (*WsfKernel)(const WsfArgs);
struct WsfForm {
  int K;
  WsfKernel k[2][2];  // [shortcut][relu]
};
#define DC_WSF_FORM(K_, D_) \
  {K_, {{ws1x1f_kernel<K_, D_, false, false>, ws1x1f_kernel<K_, D_, false, true>}, {ws1x1f_kernel<K_, D_, true, false>, ws1x1f_kernel<K_, D_, true, true>}}}
const WsfForm kFormsF[] = {DC_WSF_FORM(64, 4), DC_WSF_FORM(128, 4), DC_WSF_FORM(256, 4), DC_WSF_FORM(512, 3)};
const WsfKernel kStemWs[2] = {ws1x1f_kernel<224, 4, false, false, true>, ws1x1f_kernel<224, 4, false, true, true>};  // [relu]
const WsfForm* formf_of(int K) {
  for (const WsfForm& f : kFormsF)
    if (f.K == K) return &f;
  return nullptr;
}
void f_magic(unsigned dv, unsigned (&mg)[2]) {
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
long wsf_plan(WsfArgs& a, int klen, bool stem = false) {
  a.tn = a.Cout / 64;
  a.S = (a.M + 15) / 16;
  // one workgroup per CU; two at K = 128, where a step is 1 024 cycles of products behind the same requests and barrier (conv3_x expansion at
  // batch 1: 13.1 us with 256 workgroups, 12.1 with 512 or 768, 13.8 with 1 024; conv4_x, K = 256: 11.4 / 11.2 / 12.1 — and 2 048 waves start
  // 0.7 us later than 1 024 inside a forward)
  static const int slots_env = getenv("DC_WSF_SLOTS") ? std::max(8, atoi(getenv("DC_WSF_SLOTS"))) : 0;
  const int slots = slots_env ? slots_env : (klen <= 128 || stem) ? 512 : 256;  // (the stem: 36.7 us at 256, 32.2 at 512, 33.7 at 768; its tile 39.1)
  long J = std::min<long>(a.S, std::max(1, slots / a.tn));
  if (J >= 8) J -= J % 8;
  a.J = (int)J;
  a.sbase = a.S / a.J, a.srem = a.S % a.J;
  static const bool remap = !getenv("DC_WSF_REMAP") || atoi(getenv("DC_WSF_REMAP")) != 0;
  if (!remap) a.srem = -1;
  f_magic((unsigned)a.tn, a.div_tn);
  return (J + 7) / 8 * 8 * a.tn;
}
}  // namespace

bool stream1x1f_eligible(const ConvGemmParams& p) {
  if (p.esize != 4 || p.ncls > 1 || p.nprob > 0 || p.nty != 1 || p.ntx != 1 || p.dy0 != 0 || p.x0 != 0 || p.sy != 1 || p.sigmoid_ch != 0) return false;
  if (!formf_of(p.klen) || p.Ktot != p.klen || p.Cout % 64 != 0) return false;
  if (p.x_rows != p.OH || p.x_row_stride != p.OW * p.sx || p.x_img_stride != (long)p.OH * p.x_row_stride || p.x_rowlen < (p.OW - 1) * p.sx + p.klen) return false;
  if (p.y_row_stride != p.OW * p.y_pix_stride || p.y_img_stride != (long)p.OH * p.y_row_stride) return false;
  if ((p.sx * 4) % 16 != 0 || (p.y_pix_stride * 4) % 16 != 0) return false;
  if ((long)p.M * p.sx * 4 >= 0x7fffffffL || (long)p.M * p.y_pix_stride * 4 >= 0x7fffffffL || (long)p.M / 16 * 512 >= 0x7fffffffL) return false;
  return true;
}

long stream1x1f_grid(const ConvGemmParams& p) {
  WsfArgs a{};
  a.Cout = p.Cout, a.M = p.M;
  return wsf_plan(a, p.klen);
}

size_t stream1x1f_packed_elems(int Cout, int K) { return (size_t)Cout * K; }

// ---- the stem form: conv1 (7x7, stride 2, pad 3, <= 4 input channels as NHWC4 pixels, 64 output channels) as the lowering's row-tap launch
bool stem_ws_eligible(const ConvGemmParams& p) {
  if (p.esize != 4 || p.ncls > 1 || p.nprob > 0 || p.sigmoid_ch != 0 || p.resid) return false;
  if (p.nty != 7 || p.ntx != 1 || p.klen != 32 || p.Ktot != 224 || p.dy0 != -3 || p.ddy != 1 || p.sy != 2 || p.sx != 8 || p.x0 != -12) return false;
  if (p.Cout != 64 || p.x_rowlen % 4 != 0 || p.x_row_stride < p.x_rowlen || p.x_row_stride % 4 != 0 || p.x_img_stride % 4 != 0) return false;
  const int W = p.x_rowlen / 4, H = p.x_rows;
  if (p.OH != (H + 6 - 7) / 2 + 1 || p.OW != (W + 6 - 7) / 2 + 1 || p.M != (long)p.NB * p.OH * p.OW) return false;
  if (p.y_row_stride != p.OW * p.y_pix_stride || p.y_img_stride != (long)p.OH * p.y_row_stride || (p.y_pix_stride * 4) % 16 != 0) return false;
  if ((long)p.NB * p.x_img_stride * 4 >= 0x7fffffffL || (long)p.M * p.y_pix_stride * 4 >= 0x7fffffffL) return false;  // 32-bit byte offsets
  return true;
}

long stem_ws_grid(const ConvGemmParams& p) {
  WsfArgs a{};
  a.Cout = p.Cout, a.M = p.M;
  return wsf_plan(a, 256, true);
}

size_t stem_ws_packed_elems() { return (size_t)64 * 224; }

void stem_ws_pack_filters(const float* rowtap, float* out) { stream1x1f_pack_filters(rowtap, 64, 224, out); }  // (K = 224: 14 vectors per lane)

int launch_stem_ws(const ConvGemmParams& p, void* stream) {
  if (!stem_ws_eligible(p)) return (int)hipErrorInvalidValue;
  if (((uintptr_t)p.x & 15) || ((uintptr_t)p.y & 15) || ((uintptr_t)p.w & 15) || ((uintptr_t)p.scale & 15) || ((uintptr_t)p.shift & 15)) return (int)hipErrorInvalidValue;
  WsfArgs a{};
  a.x = p.x, a.w = p.w, a.scale = p.scale, a.shift = p.shift, a.y = p.y, a.resid = nullptr;
  a.M = p.M, a.Cout = p.Cout, a.sxb = 0, a.ypb = p.y_pix_stride * 4, a.dbg = p.dbg;
  a.OW = p.OW, a.OHW = p.OH * p.OW, a.H = p.x_rows, a.W = p.x_rowlen / 4, a.xrsb = p.x_row_stride * 4, a.ximgb = (int)(p.x_img_stride * 4);
  f_magic((unsigned)a.OW, a.div_ow);
  f_magic((unsigned)a.OHW, a.div_ohw);
  static const int abl = getenv("DC_WSF_STEM_ABL") ? atoi(getenv("DC_WSF_STEM_ABL")) : 0;  // timing only (wrong results): 1 = every request out of range (no image bytes move)
  if (abl == 1) a.H = 0;
  const long grid = wsf_plan(a, 256, true);
  if (grid <= 0 || grid > 0x7fffffffL) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(kStemWs[p.relu ? 1 : 0], dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

// g: [Cout][K] -> [Cout/16][K/16][64 lanes][4]: lane = 16 q + co % 16 holds run q of the K range (K/4 elements), 4 of them per vector:
// element e of vector j = g[co][q K/4 + 4 j + e]
void stream1x1f_pack_filters(const float* g, int Cout, int K, float* out) {
  const int NJ = K / 16;
  for (int co = 0; co < Cout; ++co)
    for (int qq = 0; qq < 4; ++qq)
      for (int j = 0; j < NJ; ++j)
        for (int e = 0; e < 4; ++e)
          out[((((size_t)(co / 16) * NJ + j) * 64) + qq * 16 + co % 16) * 4 + e] = g[(size_t)co * K + qq * (K / 4) + 4 * j + e];
}

int launch_stream1x1f(const ConvGemmParams& p, void* stream) {
  if (!stream1x1f_eligible(p)) return (int)hipErrorInvalidValue;
  if (((uintptr_t)p.x & 15) || ((uintptr_t)p.y & 15) || ((uintptr_t)p.resid & 15) || ((uintptr_t)p.w & 15) || ((uintptr_t)p.scale & 15) || ((uintptr_t)p.shift & 15))
    return (int)hipErrorInvalidValue;
  const WsfForm* f = formf_of(p.klen);
  WsfArgs a{};
  a.x = p.x, a.w = p.w, a.scale = p.scale, a.shift = p.shift, a.y = p.y, a.resid = p.resid;
  a.M = p.M, a.Cout = p.Cout, a.sxb = p.sx * 4, a.ypb = p.y_pix_stride * 4, a.dbg = p.dbg;
  const long grid = wsf_plan(a, p.klen);
  if (grid <= 0 || grid > 0x7fffffffL) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(f->k[p.resid ? 1 : 0][p.relu ? 1 : 0], dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

}  // namespace dc
