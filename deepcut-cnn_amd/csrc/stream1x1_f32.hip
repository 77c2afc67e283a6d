// stream1x1_f32.hip — the float32 form of stream1x1.hip ("ws1x1f"): the dense 1x1 expansions of a bottleneck block (+ shortcut + ReLU,
// resNx_branch2c of ResNet-152.prototxt; reference: one SGEMM per image, base_conv_layer.cpp:326-341) with the filters resident in registers
// and the pixels walked in 16-pixel steps.
//
// Why, in float32, where these layers are MFMA-bound and not byte-bound: at batch 1 (BASELINE configs[1], the headline) a conv4_x layer has
// M = 34 x 46 = 1 564 pixels = 4 x 17 x 23, and every tiling of it into 32-row MFMA fragments leaves 56-60 of the 256 CUs without a
// workgroup (196 or 200 workgroups: DESIGN.md 4.1, EXPERIMENTS.md A "why the one-forward figure sits where it does"); inside a workgroup
// 4-7 k of ~25 k cycles are prologue / split-K exchange / epilogue.  Here the unit of work is a 16-pixel x 64-channel step of
// v_mfma_f32_16x16x4_f32 (the same 64 flops per cycle and SIMD as the 32x32x2 form), the grid is ONE workgroup per CU — 16 channel
// slices x 16 pixel ranges at N = 1024 — and a workgroup walks 6 or 7 of the 98 steps of its range: 7 / 6.125 = 87.5 % of the chip's matrix
// cycles are used where 196 / 256 = 76.6 % were, the filters (64 KB per workgroup) are fetched once, and there is no split-K exchange.
//
// Per workgroup (4 waves = 4 x 16 channels; a wave alone on its SIMD):
//  * filters: registers, ROW operand (rows = 16 channels), K/4 of them per lane.  Lane (c, q) of the 16x16x4 instruction supplies
//    A[c][q] and B[q][p]: with the K range cut into four runs of K/4 — lane q takes run q, matrix step m is element m of every run — one
//    ds_read_b128 of the pixel's row feeds four matrix steps and the filters load as 16-byte vectors (stream1x1f_pack_filters);
//  * pixels: LDS-DMA ring of D stages of 16 rows (one 1 KiB request = one pixel's K floats, rows padded by 16 bytes: conflict-free reads
//    without a swizzle, so the 16 read offsets of a step are immediates), D-1 steps ahead, counted vmcnt, one barrier per step;
//  * two accumulators taken in turn (a matrix step does not wait for the one before it), added at the end;
//  * shortcut and output tiles (16 pixels x 16 channels = 1 KiB) through a wave-private LDS buffer as whole 64-byte runs, the epilogue's
//    affine / add / ReLU in fp32 in between.
// Not bit-identical to the gather-GEMM tiles (another summation grouping); same 1e-3 bound against the oracle, measured ~1e-6.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <type_traits>

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

// K = input channels (256 or 512: a pixel's row is a whole number of 1 KiB requests), D = ring stages.
// 8 waves: waves 0-3 multiply (one per SIMD, 16 channels each), waves 4-7 move bytes (the LDS-DMA requests of the ring, the shortcut tiles,
// the output stores and the shortcut add + ReLU).  A request costs the wave that issues it ~100 cycles of its instruction stream
// (MI355X_MICROARCH.md: "LDS-DMA piece issue cost") and the partner on its SIMD ~20: in the multiplying wave's own stream five requests per
// 16-pixel step were a quarter of the step (the first form of this kernel: 2 600 cycles per 2 048 cycles of products).
template <int K, int D, bool RES, bool RELU, int ABL = 0>
__global__ __launch_bounds__(512, 1) void ws1x1f_kernel(const WsfArgs a) {
  const long long t_entry = (long long)__builtin_amdgcn_s_memrealtime();
  constexpr int ROWB = K * 4 + 16;   // bytes of a pixel's row in LDS (padded: lanes p = 0..15 of a read land in 16 different bank groups)
  constexpr int STG = 16 * ROWB;     // a stage: 16 pixels
  constexpr int PPR = K * 4 / 1024;  // 1 KiB requests per pixel row
  constexpr int NA = 16 * PPR / 4;   // ... per moving wave and stage
  constexpr int NJ = K / 16;         // 16-byte reads (= 4 matrix steps each) per multiplying wave and step
  constexpr int NR = RES ? 1 : 0, NS = 1;
  static_assert(K % 256 == 0 && D >= 3 && D <= 4, "row = whole requests; the ring");
  using Rq = ReqCount<D, NA, NR, NS>;
  constexpr int OB0 = D * STG;             // 4 waves x 2 x 1 KiB: the product tiles (after scale / shift), written in the matrix view
  constexpr int RB0 = OB0 + 4 * 2 * 1024;  // 4 waves x 2 x 1 KiB: the shortcut tiles, in the memory view
  constexpr int LDSB = RB0 + 4 * 2 * 1024;
  static_assert(LDSB <= 160 * 1024, "LDS of a CU");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[LDSB];
  const int t = threadIdx.x, lane = t & 63, wave8 = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wave = wave8 & 3;
  const bool mover = wave8 >= 4 && ABL != 4 && ABL != 5;  // (4 / 5: all eight waves multiply — twice the products, no data)
  const int p16 = lane & 15, q = lane >> 4;

  const int bx = blockIdx.x, xcd = bx & 7, bi = bx >> 3;
  const int ci = f_fastdiv(bi, a.div_tn);
  const int nt = bi - ci * a.tn;
  const int jc = xcd + 8 * ci;
  if (jc >= a.J) return;
  int gs0 = jc * a.sbase + min(jc, a.srem), gs1 = gs0 + a.sbase + (jc < a.srem ? 1 : 0);
  if (a.srem < 0) gs0 = a.S * jc / a.J, gs1 = a.S * (jc + 1) / a.J;  // (DC_WSF_REMAP=0: the longer ranges spread over the grid, for A/B timing)
  const int nw0 = nt * 64 + wave * 16;  // first channel of this wave (of the pair on this SIMD)
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  auto stamp = [&](int sl) {
    if (a.dbg && lane == 0 && !mover) {
      long long* d = a.dbg + ((long)blockIdx.x * 4 + wave) * 12;
      d[sl] = (long long)__builtin_readcyclecounter();
      if (sl == 0) d[8] = t_entry, d[10] = (long long)__builtin_amdgcn_s_memrealtime();
      if (sl == 7) d[9] = (long long)__builtin_amdgcn_s_memrealtime();
    }
  };

  if (mover) {
    // ================================================================ the moving wave
    const i32x4 xr = f_rsrc_words(a.x), rr = f_rsrc_words(RES ? a.resid : a.y), yr = f_rsrc_words(a.y);
    const int orow = lane >> 2;  // memory view: pixel lane / 4, channels 4 (lane % 4) .. + 3 of the wave's 16
    const unsigned ooff = (unsigned)((nw0 + 4 * (lane & 3)) * 4);
    auto dma_req = [&](int gs, int slot, int i) {  // request i of the wave's NA for stage gs (issued whatever gs is: beyond the range it moves nothing)
      const int row0 = gs * 16;
      const int lim = gs < gs1 ? a.M - row0 : 0;
      const int pc = wave + 4 * i, row = pc / PPR, part = pc - row * PPR;  // request pc of the stage: part `part` of row `row`
      const unsigned vo = row < lim ? (unsigned)(row0 + row) * (unsigned)a.sxb + (unsigned)(part * 1024 + lane * 16) : kOOBf;
      f_dma16(xr, lds0 + (unsigned)(slot * STG + row * ROWB + part * 1024), vo);
    };
    auto resid_req = [&](int gs, int buf) {  // NR requests, always
      if (!RES) return;
      const int row0 = gs * 16;
      const int lim = gs < gs1 ? a.M - row0 : 0;
      f_dma16(rr, lds0 + (unsigned)(RB0 + (wave * 2 + buf) * 1024), orow < lim ? (unsigned)(row0 + orow) * (unsigned)a.ypb + ooff : kOOBf);
    };
    auto finish = [&](int gs, int buf) {  // product tile + shortcut tile of step gs -> memory
      const f32x4 o = *reinterpret_cast<const f32x4*>(smem + OB0 + (wave * 2 + buf) * 1024 + lane * 16);
      f32x4 rv = {0.f, 0.f, 0.f, 0.f};
      if (RES) rv = *reinterpret_cast<const f32x4*>(smem + RB0 + (wave * 2 + buf) * 1024 + lane * 16);
      u32x4 ov;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = o[r];
        if (RES) v += rv[r];
        ov[r] = __float_as_uint(RELU ? fmaxf(v, 0.f) : v);
      }
      const int row0 = gs * 16;
      f_store16_untracked(yr, orow < a.M - row0 ? (unsigned)(row0 + orow) * (unsigned)a.ypb + ooff : kOOBf, ov);
    };
#pragma unroll
    for (int d = 0; d < D - 1; ++d)
#pragma unroll
      for (int i = 0; i < NA; ++i) dma_req(gs0 + d, d, i);
    resid_req(gs0, 0);
    resid_req(gs0 + 1, 1);
    f_wait_vm<Rq::pro>();
    f_lds_barrier();
    int slot = 0, ob = 0;
    auto do_step = [&](int gs, auto k_tag) {
      constexpr int KS = decltype(k_tag)::value;  // the step's number while the request counts still change (Rq), -1 in the steady state
      constexpr int KQ = KS < 0 ? Rq::kSteady : KS;
      const int nslot = slot + 1 == D ? 0 : slot + 1, pslot = slot == 0 ? D - 1 : slot - 1;
      f_wait_vm<Rq::stage(KQ)>();
      if (ABL != 2 && ABL != 5) f_lds_barrier();  // stage gs+1 is whole; the slot of step gs-1 is free; the product tile of step gs-1 is written
#pragma unroll
      for (int i = 0; i < NA; ++i) dma_req(gs + D - 1, pslot, i);
      if (KS != 0) {
        if (RES) f_wait_vm<Rq::resid(KQ)>();
        finish(gs - 1, ob);
        resid_req(gs + 1, ob);
        ob ^= 1;
      }
      slot = nslot;
    };
    int gs = gs0;
    if (gs < gs1) do_step(gs++, std::integral_constant<int, 0>{});
    if (gs < gs1) do_step(gs++, std::integral_constant<int, 1>{});
    if (gs < gs1) do_step(gs++, std::integral_constant<int, 2>{});
    static_assert(Rq::kPeel == 3, "steps 0, 1, 2 above");
    for (; gs < gs1; ++gs) do_step(gs, std::integral_constant<int, -1>{});
    f_wait_vm<0>();
    f_lds_barrier();  // the last product tile is written
    finish(gs1 - 1, ob);  // (ob names the last step's tile: it has flipped once per finished step)
    return;
  }

  // ================================================================ the multiplying wave
  // ---- epilogue constants (4 channels per lane: 4 q + r) and filters: the wave's only vector-memory requests
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
  const unsigned frag0 = (unsigned)(p16 * ROWB + q * K);  // this lane's run of the pixel's row: + 16 j
  const unsigned ovec = (unsigned)((4 * p16 + q) * 16);   // matrix view: pixel p16, channels 4 q .. + 3
  stamp(0);
  stamp(1);
  f_wait_vm<0>();
  asm volatile("" : "+v"(csc), "+v"(csh));
#pragma unroll
  for (int j = 0; j < NJ; ++j) asm volatile("" : "+v"(wreg[j]));
  f_lds_barrier();  // stage 0 is whole
  stamp(2);

  // One step = the NJ x 4 matrix products of 16 pixels.  The matrix pipe runs from the first step's first product to the last step's last
  // one: the operand reads run PD-1 reads ahead across the step boundary, and the step's one barrier (stage k+1 published, the tile of
  // step k-1 handed to the moving wave) sits between its products.
  constexpr int PD = 4;
  static_assert(NJ % PD == 0, "the operand ring keeps its phase across steps");
  int slot = 0, ob = 0;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f}, acc3 = {0.f, 0.f, 0.f, 0.f};
  f32x4 xf[PD];
#pragma unroll
  for (int j = 0; j < PD - 1; ++j) xf[j] = *reinterpret_cast<const f32x4*>(smem + frag0 + j * 16);
  auto hand_over = [&](int buf) {  // register r of a lane = channel 4 q + r of pixel p16
    const f32x4 sum = ABL == 3 ? (acc0 + acc1) + (acc2 + acc3) : acc0 + acc1;
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = sum[r] * csc[r] + csh[r];
    *reinterpret_cast<f32x4*>(smem + OB0 + (wave * 2 + buf) * 1024 + ovec) = o;
  };
  auto do_step = [&](bool first) {
    constexpr int HB = 1;
    static_assert(HB < NJ - PD + 1, "stage k+1 is read only behind its barrier");
    const int nslot = slot + 1 == D ? 0 : slot + 1;
    const unsigned char* st = smem + slot * STG + frag0;
    const unsigned char* stn = smem + nslot * STG + frag0;
    if (!first) {
      hand_over(ob);
      ob ^= 1;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int jn = j + PD - 1;
      if (ABL != 1 && ABL != 2 && ABL != 5) xf[jn % PD] = *reinterpret_cast<const f32x4*>(jn < NJ ? st + jn * 16 : stn + (jn - NJ) * 16);
      if (ABL == 3) {  // four accumulators
        if (j == 0) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[j][0], xf[j % PD][0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[j][1], xf[j % PD][1], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[j][2], xf[j % PD][2], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[j][3], xf[j % PD][3], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        } else {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[j][0], xf[j % PD][0], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[j][1], xf[j % PD][1], acc1, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[j][2], xf[j % PD][2], acc2, 0, 0, 0);
          acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[j][3], xf[j % PD][3], acc3, 0, 0, 0);
        }
      } else {
      if (j == 0) {  // fresh accumulators: the first two products take the constant 0
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[j][0], xf[j % PD][0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[j][1], xf[j % PD][1], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      } else {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[j][0], xf[j % PD][0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[j][1], xf[j % PD][1], acc1, 0, 0, 0);
      }
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[j][2], xf[j % PD][2], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[j][3], xf[j % PD][3], acc1, 0, 0, 0);
      }
      if (j == HB && ABL != 2 && ABL != 5) f_lds_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
    slot = nslot;
  };
  int gs = gs0;
  if (gs < gs1) do_step(true), ++gs;
  stamp(3);
  for (; gs < gs1; ++gs) do_step(false);
  stamp(4);
  hand_over(ob);
  stamp(5);
  f_lds_barrier();
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
const WsfForm kFormsF[] = {DC_WSF_FORM(256, 4), DC_WSF_FORM(512, 3)};
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
long wsf_plan(WsfArgs& a) {
  a.tn = a.Cout / 64;
  a.S = (a.M + 15) / 16;
  static const int slots = getenv("DC_WSF_SLOTS") ? std::max(8, atoi(getenv("DC_WSF_SLOTS"))) : 256;  // one workgroup per CU
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
  return wsf_plan(a);
}

size_t stream1x1f_packed_elems(int Cout, int K) { return (size_t)Cout * K; }

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
  const long grid = wsf_plan(a);
  if (grid <= 0 || grid > 0x7fffffffL) return (int)hipErrorInvalidValue;
  WsfKernel kern = f->k[p.resid ? 1 : 0][p.relu ? 1 : 0];
  static const int abl = getenv("DC_WSF_ABL") ? atoi(getenv("DC_WSF_ABL")) : 0;  // timing ablations of the conv4_x form (wrong results): tools/stream1x1_probe.py
  if (abl && p.klen == 256 && p.resid && p.relu) {
    // 1: no operand reads in the loop, 2: neither reads nor barriers, 3: four accumulators instead of two (right results)
    // 4: all eight waves multiply (two per SIMD, twice the products, no data moved), 5: 4 without reads and barriers
    static const WsfKernel kAbl[6] = {nullptr, ws1x1f_kernel<256, 4, true, true, 1>, ws1x1f_kernel<256, 4, true, true, 2>, ws1x1f_kernel<256, 4, true, true, 3>,
                                      ws1x1f_kernel<256, 4, true, true, 4>, ws1x1f_kernel<256, 4, true, true, 5>};
    if (abl >= 1 && abl <= 5) kern = kAbl[abl];
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

}  // namespace dc
