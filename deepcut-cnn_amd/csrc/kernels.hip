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
#include <cstdlib>
#include <cstring>
#include <type_traits>

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
// n / d for 0 <= n < 2^31 with host-computed magic {mul, shift}: 2 VALU instead of the ~25 of a runtime division
__device__ __forceinline__ int dc_fastdiv(int n, const unsigned (&mg)[2]) {
  return (mg[1] >> 31) ? n : (int)(__umulhi((unsigned)n, mg[0]) >> (mg[1] & 31));  // bit 31 of the shift word: d == 1
}

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// LDS-DMA (`buffer_load_dwordx4 ... lds`): 64 lanes x 16 bytes travel from global memory straight into LDS, no VGPRs and
// no ds_write.  The LDS destination is M0 + 16*lane (lane-linear, 1 KiB per wave instruction); the SOURCE address is per
// lane (V# base + voffset + soffset), so a swizzled LDS image is made by permuting which 16-byte chunk each lane
// fetches.  An out-of-range voffset stores zeros (the zero padding of the gather keeps working unchanged).
// Written as inline asm on purpose: through the builtin the compiler knows that LDS is written behind its back and
// makes every later ds_read wait for vmcnt(0) — the pipeline below keeps 1-2 tiles in flight across its barriers and
// counts vmcnt itself.  (M0 is written in the same statement that reads it; the compiler does not use M0 on this path.)
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ i32x4 dc_rsrc_words(const void* p) {
  const unsigned long long a = (unsigned long long)p;
  return i32x4{(int)(unsigned)a, (int)((a >> 32) & 0xffffu), 0x7fffffff, 0x00020000};
}
__device__ __forceinline__ void dc_dma16(i32x4 rs, unsigned lds, unsigned voff, unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds), "v"(voff), "s"(rs), "s"(soff)
               : "memory", "m0");
}
// a 4-byte buffer load the compiler does not track (no s_waitcnt of its own): the caller's counted vmcnt covers it
__device__ __forceinline__ float dc_load_f32_untracked(i32x4 rs, unsigned voff) {
  float v;
  // s_nop: the hazard recogniser does not look inside inline asm, and "VALU writes SGPR -> VMEM reads that SGPR" needs 5
  // wait states (a descriptor restored from an SGPR spill by v_readlane right in front of this statement read stale
  // registers in the round-3 walking-tile experiment: wild addresses).  tools/check_asm_hazards.py scans for the pattern.
  asm volatile("s_nop 4\n\tbuffer_load_dword %0, %1, %2, 0 offen" : "=v"(v) : "v"(voff), "s"(rs) : "memory");
  return v;
}
// lanes 32..63 of lo[e] <-> lanes 0..31 of hi[e], e = 0..3 (inline asm: this compiler's builtin returns the first result
// twice; one s_nop for the four: the VALU instructions that produced the operands need two wait states before a permlane)
__device__ __forceinline__ void dc_permlane32_swap4(float (&lo)[4], float (&hi)[4]) {
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %4\n\tv_permlane32_swap_b32 %1, %5\n\tv_permlane32_swap_b32 %2, %6\n\tv_permlane32_swap_b32 %3, %7"
      : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]));
}
// float32 + the low / high half of a packed float16 pair, exactly rounded once (v_fma_mix_f32 h * 1.0 + f): the conversion folded into
// the add.  Plain (non-volatile) asm: pure functions of their inputs.
__device__ __forceinline__ float dc_add_half_lo(unsigned h2, float f) {
  float d;
  asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h2), "v"(f));
  return d;
}
__device__ __forceinline__ float dc_add_half_hi(unsigned h2, float f) {
  float d;
  asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h2), "v"(f));
  return d;
}
template <int N>
__device__ __forceinline__ void dc_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// own LDS traffic retired, then the workgroup barrier (a raw s_barrier: __syncthreads() would drain vmcnt too)
__device__ __forceinline__ void dc_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Kernel arguments are fetched lazily, cache line by cache line, wherever the compiler first needs a field: the gather-GEMM
// touched four lines of its 632-byte block at four different points of its prologue, each first touch a scalar-cache miss
// on the critical path.  DC_KARG_TOUCH requests one dword of each of the first five lines (every field but the multi-class table) at kernel entry (results unused), so
// that the lines arrive together; DC_KARG_HOLD keeps the five scratch SGPRs reserved until a point that the compiler can
// only reach after an `s_waitcnt lgkmcnt(0)` of its own (scalar loads return out of order: a register freed earlier could
// be overwritten by the late dummy load).
#define DC_KARG_TOUCH(n0, n1, n2, n3, n4)                                                                                  \
  unsigned n0, n1, n2, n3, n4;                                                                                             \
  {                                                                                                                        \
    const auto* ka_ = __builtin_amdgcn_kernarg_segment_ptr();                                                              \
    asm volatile("s_load_dword %0, %5, 0x0\n\ts_load_dword %1, %5, 0x40\n\ts_load_dword %2, %5, 0x80\n\ts_load_dword %3, %5, 0xc0\n\t" \
                 "s_load_dword %4, %5, 0x100"                                                                              \
                 : "=&s"(n0), "=&s"(n1), "=&s"(n2), "=&s"(n3), "=&s"(n4)                                                     \
                 : "s"(ka_));                                                                                              \
  }
#define DC_KARG_HOLD(n0, n1, n2, n3, n4) asm volatile("" ::"s"(n0), "s"(n1), "s"(n2), "s"(n3), "s"(n4))
static_assert(sizeof(ConvGemmParams) >= 0x140, "DC_KARG_TOUCH reads five 64-byte lines of the argument block");

// element-type traits of the gather-GEMM: activations / filters are float or _Float16 in HBM and LDS,
// accumulation, the epilogue arithmetic and its constants are always float
template <typename T>
struct Elem;
template <>
struct Elem<float> {
  static constexpr int SPC = 4;  // MFMA steps per 32-byte operand chunk: 4 x v_mfma_f32_32x32x2_f32 (k = 8)
  typedef unsigned raw_t;  // a loaded element before conversion (conversion = first use: the load stays in flight until then)
  static __device__ __forceinline__ raw_t load_raw(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0); }
  static __device__ __forceinline__ float cvt(raw_t v) { return __builtin_bit_cast(float, v); }
  static __device__ __forceinline__ float load(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
  }
  static __device__ __forceinline__ void store(float v, __amdgpu_buffer_rsrc_t r, unsigned off) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, off, 0, 0);
  }
};
template <>
struct Elem<_Float16> {
  static constexpr int SPC = 1;  // 1 x v_mfma_f32_32x32x16_f16 (k = 16) per 32-byte chunk
  typedef unsigned short raw_t;
  static __device__ __forceinline__ raw_t load_raw(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b16(r, off, 0, 0); }
  static __device__ __forceinline__ float cvt(raw_t v) { return (float)__builtin_bit_cast(_Float16, v); }
  static __device__ __forceinline__ float load(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return (float)__builtin_bit_cast(_Float16, __builtin_amdgcn_raw_buffer_load_b16(r, off, 0, 0));
  }
  static __device__ __forceinline__ void store(float v, __amdgpu_buffer_rsrc_t r, unsigned off) {
    __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (_Float16)v), r, off, 0, 0);
  }
};

// MC: multi-class launch (ConvGemmParams::ncls > 1): the class-dependent scalars come from p.cls[class of this block].
// DMA: 0 = operands staged through a register ring (PF tiles) and ds_write_b128 into two padded LDS stages;
//      D >= 2 = operands fetched by LDS-DMA into a ring of D unpadded, XOR-swizzled LDS stages (tile rows of 128 bytes),
//      D - 1 tiles requested ahead, one raw barrier per tile, vmcnt counted by hand (PF unused).
// SWP: the MFMA operands are swapped (filters as the row operand), so that a lane's accumulators are 16 CHANNELS of one
//      pixel (4 runs of 4 consecutive channels) instead of 16 pixels of one channel: the epilogue then forms 16-byte output
//      vectors in registers (float16: one v_permlane32_swap per register pair) — no LDS transposition, no barriers.
// MP: multi-problem launch (ConvGemmParams::nprob > 0): every per-tensor scalar AND pointer comes from the device-resident
//     table p.multi->prob[problem of this block] — the same layer over the scales of a pyramid / the classes of a deconvolution.
template <typename T, int BM, int BN, int BK, int WR, int WC, int WK, int PF, bool MC = false, int DMA = 0, bool SWP = false, bool MP = false>
// (second launch bound = waves per SIMD the register budget must allow: the 4-wave LDS-DMA tiles are meant to run two
//  workgroups per CU, so their allocation has to stay within 256 registers — with it the compiler also keeps the
//  accumulators in VGPRs instead of AGPRs: no v_accvgpr_read pass in front of the epilogue, 169 instead of 200 registers)
__global__ __launch_bounds__(WR* WC* WK * 64, (DMA && WR * WC * WK == 4) ? 2 : 1) void conv_gemm_kernel(const std::conditional_t<MP, ConvMultiArgs, ConvGemmParams> ka) {
  const ConvGemmParams& p = [&]() -> const ConvGemmParams& {
    if constexpr (MP) return ka.p;
    else return ka;
  }();
  const long long t_entry = (long long)__builtin_amdgcn_s_memrealtime();  // before the first kernel-argument load (DC_DEBUG_TIMING)
  DC_KARG_TOUCH(ka0, ka1, ka2, ka3, ka4);
  constexpr int ES = sizeof(T);          // bytes per element
  constexpr int VEC = 16 / ES;           // elements per 16-byte vector
  constexpr int SPC = Elem<T>::SPC;
  constexpr int NW = WR * WC * WK;       // waves per workgroup (4 or 8)
  constexpr int NT = NW * 64;
  static_assert(NW == 4 || NW == 8 || (NW == 16 && DMA), "4 or 8 waves per workgroup (16 compiles for the LDS-DMA tiles: measured, no gain — EXPERIMENTS.md A)");
  constexpr int LDB = DMA ? BK * ES : BK * ES + 16;  // LDS row, bytes: padded (16-B aligned, bank-spread), or 128 swizzled (DMA)
  static_assert(!DMA || BK * ES == 128 || BK * ES == 256, "the LDS-DMA image has 128- or 256-byte rows (8 / 16 chunks of 16 bytes, XOR-swizzled)");
  static_assert(DMA == 0 || (DMA >= 2 && DMA <= 4), "ring of 2..4 LDS stages");
  constexpr int NSTG = DMA ? DMA : 2;
#ifndef DC_PRO_FULL
#define DC_PRO_FULL 1
#endif
  // LDS-DMA: the prologue requests the first DMA-1 tiles (all their filter pieces before the row decode, their activation pieces
  // after it) and then — LATE — the whole of tile DMA-1, which the steady state (tile `it` requests tile it+DMA-1) would only ask for
  // during tile 0: a first-tile latency earlier, without putting another tile's filter pieces in front of tile 0's activations
  // (requested with the first filter pieces instead, it cost the float32 batch-1 forward 1.4 %).
  // float16 only: the float32 batch-1 launches are one round of workgroups that all start together, and a deeper initial burst
  // delays their first tile more than the earlier tile DMA-1 gives back (one forward at a time 335.8 -> 331.5 images/s).
  constexpr bool LATE = DMA && DC_PRO_FULL && ES == 2;
  constexpr int TM = BM / WR, TN = BN / WC;
  constexpr int FM = TM / 32, FN = TN / 32;
  static_assert(FM >= 1 && FN >= 1 && TM % 32 == 0 && TN % 32 == 0, "wave tile = multiples of 32x32");
  constexpr int C4 = BK / VEC;           // 16-byte vectors per tile row
  constexpr int RPP = NT / C4;           // tile rows covered by one pass of the workgroup's threads
  constexpr int NA = BM / RPP, NBV = BN / RPP;
  static_assert(NA >= 1 && NBV >= 1, "tile too small for the loader");
  constexpr int KCH = BK * ES / 32;      // 32-byte operand chunks per tile row
  static_assert(KCH % WK == 0, "k chunks must split evenly over WK");
  constexpr int NCH = KCH / WK;          // chunks this wave owns per tile
  static_assert(NCH >= 2 && NCH % 2 == 0, "the software pipeline needs an even number (>=2) of chunks per wave");
  constexpr int NST = NCH * SPC;         // MFMA steps per tile per wave
  static_assert(DMA || NST >= 4, "the register-ring pipeline places its work in 4 distinct MFMA steps");
  constexpr int GBAR = SPC > 1 ? NST - 3 : NST - 1;  // step before which the tile's barrier is taken
  constexpr int TILEB = (BM + BN) * LDB;  // bytes per LDS stage
  constexpr int RPW = 16 / WK;           // accumulator registers each split-K wave finalises
  constexpr int WPS = BN + 4;                       // wide-epilogue staging row pitch (floats)
  // DMA: the buffer is sized for the ring, the split-K exchange and the wide epilogue's staging, whichever is largest
  constexpr int EPIB = ES == 2 && !SWP ? BM * WPS * 4 : 0;
  constexpr int SPKB = (WK - 1) * BM * BN * 4;
  constexpr int MAINB = DMA ? (NSTG * TILEB > EPIB ? (NSTG * TILEB > SPKB ? NSTG * TILEB : SPKB) : (EPIB > SPKB ? EPIB : SPKB)) : 2 * TILEB;
  static_assert(SPKB <= MAINB, "split-K partials must fit in the tile buffers");
  constexpr bool EARLY_RESID = !SWP && FM * FN * RPW <= 16;  // shortcut tile prefetched before the K loop
  // Wide epilogue (float16): the C fragment gives a lane ONE channel of 16 different pixels, i.e. 2-byte accesses — 128 vector
  // memory instructions per lane for a 64x64 wave tile with a shortcut, and the CU's address unit, not HBM, bounds the
  // bandwidth-bound layers (K <= 512).  Instead the finished tile (fp32, after the affine) is transposed through the tile
  // buffers in LDS and every thread adds the shortcut to, and stores, 8 consecutive channels of one pixel: 16-byte accesses,
  // 8x fewer instructions; the shortcut vectors are requested before the transposition so that they are in flight meanwhile.
  constexpr int WVEC = BM * (BN / 8) / NT;           // 16-byte output vectors per thread
  constexpr bool WIDE_OK = !SWP && ES == 2 && BM * WPS * 4 <= MAINB && (BM * (BN / 8)) % NT == 0 && WVEC <= 16;

  static_assert(!SWP || (DMA && (16 / WK) % 4 == 0), "SWP: LDS-DMA kernels, every split-K wave finalises whole 4-channel runs");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[MAINB + 4 * BM + (SWP ? 8 * BN : 0)];
  int* rowinfo = reinterpret_cast<int*>(smem + MAINB);
  float* epi_sc = reinterpret_cast<float*>(smem + MAINB + 4 * BM);  // SWP: scale[BN], shift[BN] of the tile's channels
  // per tile row: byte offset of its output pixel, or -1
  static_assert(!(MC && MP), "a multi-problem launch carries its deconvolution classes as problems");

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = t >> 6;
  const int wk = wave % WK;
  const int wc = (wave / WK) % WC;
  const int wr = wave / (WK * WC);

  // Tile <- block map.  Blocks are observed to land on XCD (blockIdx % 8), each XCD with its own 4 MB L2.
  // The default n-fastest order makes every XCD touch every filter row AND every pixel row; instead the
  // tile grid is cut into 8 rectangles (gx x gy chosen on the host to minimise filters*gy + pixels*gx)
  // and XCD q walks rectangle q, so an L2 only fetches its rectangle's share of both operands.
  // Correctness does not depend on the placement (it is only a locality hint).
  // class-dependent scalars (uniform: SGPRs).  Single-problem launches read them from the parameter block itself.
  int c_Ktot = p.Ktot, c_nty = p.nty, c_ntx = p.ntx, c_dy0 = p.dy0, c_ddy = p.ddy, c_x0 = p.x0, c_ddx = p.ddx, c_xbias = p.x_bias;
  int c_OH = p.OH, c_OW = p.OW, c_M = p.M;
  unsigned c_dohw[2] = {p.div_ohw[0], p.div_ohw[1]}, c_dow[2] = {p.div_ow[0], p.div_ow[1]};
  long c_woff = 0, c_yoff = 0;
  // per-tensor scalars (uniform): the argument block's in a single-problem launch, the problem's in a multi-problem one
  const T* px = reinterpret_cast<const T*>(p.x);
  long c_ximg = p.x_img_stride, c_yimg = p.y_img_stride;
  int c_xrow = p.x_row_stride, c_xrows = p.x_rows, c_xrowlen = p.x_rowlen, c_yrow = p.y_row_stride, c_ypix = p.y_pix_stride;
  void* c_y = p.y;
  const void* c_resid = p.resid;
  int c_densex = p.dense_x, c_densey = p.dense_y;
  int tile_n, tile_m;
  if constexpr (MP) {
    // XCD (qx, qy) owns the n tiles [tn*qx/gx, tn*(qx+1)/gx) and walks, problem after problem, the m tiles
    // [tm_k*qy/gy, tm_k*(qy+1)/gy) of every problem k, n fastest: the layer's filters (shared by all problems) are fetched
    // once per L2 that needs them.  The grid is 8 x the longest XCD list; surplus workgroups exit.
    const ConvMultiTable* tb = &ka.t;  // in the argument block: scalar loads
    const int xq = blockIdx.x & 7, lgx = p.mc_lgx, lgy = 3 - lgx;
    const int qx = xq & ((1 << lgx) - 1), qy = xq >> lgx;
    const int n_lo = (p.tiles_n * qx) >> lgx, n_cnt = ((p.tiles_n * (qx + 1)) >> lgx) - n_lo;
    const int slot = blockIdx.x >> 3;
    const int mrow = n_cnt == (p.tiles_n >> lgx) ? dc_fastdiv(slot, p.div_rw[0]) : dc_fastdiv(slot, p.div_rw[1]);  // slot / n_cnt
    int k = 0, start = 0;
#pragma unroll
    for (int j = 0; j < kMaxProblems; ++j) {
      const int e = tb->end[qy][j];
      const int ge = (e - 1 - mrow) >> 31;  // -1 iff mrow >= e (sign arithmetic: stays on the scalar unit; e <= INT_MAX, mrow >= 0)
      k -= ge;
      start = (e & ge) | (start & ~ge);
    }
    if (k >= p.nprob) return;
    const ConvProblem& q = tb->prob[k];
    c_Ktot = q.Ktot, c_nty = q.nty, c_ntx = q.ntx, c_dy0 = q.dy0, c_ddy = q.ddy, c_x0 = q.x0, c_ddx = q.ddx, c_xbias = q.x_bias;
    c_OH = q.OH, c_OW = q.OW, c_M = q.M;
    c_dohw[0] = q.div_ohw[0], c_dohw[1] = q.div_ohw[1], c_dow[0] = q.div_ow[0], c_dow[1] = q.div_ow[1];
    c_woff = q.w_off;
    px = reinterpret_cast<const T*>(q.x);
    c_ximg = q.x_img_stride, c_yimg = q.y_img_stride;
    c_xrow = q.x_row_stride, c_xrows = q.x_rows, c_xrowlen = q.x_rowlen, c_yrow = q.y_row_stride, c_ypix = q.y_pix_stride;
    c_y = q.y;
    c_resid = q.resid;
    c_densex = q.dense_x, c_densey = q.dense_y;
    tile_m = ((q.tiles_m * qy) >> lgy) + (mrow - start);
    tile_n = n_lo + (slot - mrow * n_cnt);
  } else if constexpr (MC) {
    // Workgroup b runs on XCD (b % 8), each with its own L2.  XCD (qx, qy) of a gx x gy arrangement owns the n tiles
    // [tn*qx/gx, tn*(qx+1)/gx) and, of EVERY class, the m tiles [tm*qy/gy, tm*(qy+1)/gy); it walks its classes heaviest
    // first.  So an L2 fetches 1/gx of the filter images and 1/gy of the pixels instead of everything (the heads'
    // filters alone are 30 MB against 4 MB of L2).  The grid is 8 x the longest XCD list; surplus workgroups exit.
    const int xq = blockIdx.x & 7, lgx = p.mc_lgx, lgy = 3 - lgx;
    const int qx = xq & ((1 << lgx) - 1), qy = xq >> lgx;
    const int n_lo = (p.tiles_n * qx) >> lgx, n_cnt = ((p.tiles_n * (qx + 1)) >> lgx) - n_lo;
    int slot = blockIdx.x >> 3, c = -1, m_lo = 0;
#pragma unroll
    for (int k = 0; k < kMaxClasses; ++k)
      if (k < p.ncls && c < 0) {
        const int lo = (p.cls[k].tiles_m * qy) >> lgy;
        const int cnt = (((p.cls[k].tiles_m * (qy + 1)) >> lgy) - lo) * n_cnt;
        if (slot < cnt) c = k, m_lo = lo;
        else slot -= cnt;
      }
    if (c < 0) return;
    const ConvClass& q = p.cls[c];
    c_Ktot = q.Ktot, c_nty = q.nty, c_ntx = q.ntx, c_dy0 = q.dy0, c_ddy = q.ddy, c_x0 = q.x0, c_ddx = q.ddx, c_xbias = q.x_bias;
    c_OH = q.OH, c_OW = q.OW, c_M = q.M;
    c_dohw[0] = q.div_ohw[0], c_dohw[1] = q.div_ohw[1], c_dow[0] = q.div_ow[0], c_dow[1] = q.div_ow[1];
    c_woff = q.w_off, c_yoff = q.y_off;
    const int sr = slot / n_cnt;  // uniform
    tile_m = m_lo + sr;
    tile_n = n_lo + (slot - sr * n_cnt);
  } else if (p.xcd_on) {
    const int q = blockIdx.x & 7, slot = blockIdx.x >> 3;  // XCD, position inside its rectangle
    const int lgx = p.xcd_lgx, lgy = 3 - lgx, qx = q & ((1 << lgx) - 1), qy = q >> lgx;
    const int n_lo = (p.tiles_n * qx) >> lgx, rw = ((p.tiles_n * (qx + 1)) >> lgx) - n_lo;
    const int m_lo = (p.tiles_m * qy) >> lgy, rh = ((p.tiles_m * (qy + 1)) >> lgy) - m_lo;
    if (slot >= rw * rh) return;  // grid is padded to 8 x the largest rectangle
    const int sr = rw == (p.tiles_n >> lgx) ? dc_fastdiv(slot, p.div_rw[0]) : dc_fastdiv(slot, p.div_rw[1]);  // slot / rw, all scalar
    tile_n = n_lo + (slot - sr * rw);
    tile_m = m_lo + sr;
  } else {
    tile_m = dc_fastdiv(blockIdx.x, p.div_tn);
    tile_n = blockIdx.x - tile_m * p.tiles_n;
  }
  const int m0 = tile_m * BM;
  const int n0 = tile_n * BN;
  const int T_ = c_Ktot / BK;
  DC_KARG_HOLD(ka0, ka1, ka2, ka3, ka4);  // the tile map above needed kernel arguments: the dummy loads have landed
  auto stamp = [&](int slot) {
    if (p.dbg && lane == 0) {
      long long* d = p.dbg + ((long)blockIdx.x * NW + wave) * 12;
      d[slot] = (long long)__builtin_readcyclecounter();
      if (slot == 0) d[8] = t_entry, d[10] = (long long)__builtin_amdgcn_s_memrealtime();  // 100 MHz, the same clock on every CU
      if (slot == 7) d[9] = (long long)__builtin_amdgcn_s_memrealtime();
    }
  };
  stamp(0);

  // ---- prologue, ordered so that every exposed memory round trip overlaps another -------------------
  // (1) filter rows need no pixel decode: their first PF tiles go out immediately
  const int lrow = t / C4;
  // byte column of this thread's 16-byte vector.  DMA: lane l of a wave lands at LDS byte 16*l of its 1 KiB piece (8 rows of
  // 128 bytes: row l/8, position l%8 — or 4 rows of 256 bytes), and position p of row r holds source chunk p ^ swz(r),
  // swz(r) = (r >> 1) & 7 for 128-byte rows, r & 15 for 256-byte rows: the 16 lanes a ds_read_b128 serves together (rows
  // {0-3,12-15,20-27} / {4-11,16-19,28-31} of a fragment, same chunk) then hit 16 different 16-byte bank groups, without padding.
  constexpr bool ROW256 = BK * ES == 256;
  const int lcb = DMA ? ((t % C4) ^ (ROW256 ? (lrow & 15) : ((lrow >> 1) & 7))) * 16 : (t % C4) * 16;
  const int lce = lcb / ES;       // same, in elements
  static_assert(!DMA || (C4 == (ROW256 ? 16 : 8) && RPP % (ROW256 ? 16 : 32) == 0), "DMA loader: the swizzle of a row must not depend on the piece set");
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const unsigned ldsw = lds0 + (unsigned)__builtin_amdgcn_readfirstlane(wave) * 1024u;  // this wave's piece of every 8*NW-row set
  unsigned bvoff[NBV];
#pragma unroll
  for (int j = 0; j < NBV; ++j) {
    const int n = n0 + lrow + RPP * j;
    bvoff[j] = n < p.Cout ? (unsigned)(n * c_Ktot + lce) * ES : kOOB;  // rows past Cout read as zeros
  }
  const __amdgpu_buffer_rsrc_t wr_ = dc_rsrc(reinterpret_cast<const T*>(p.w) + c_woff, 0x7fffffffu);
  const i32x4 wrs = dc_rsrc_words(reinterpret_cast<const T*>(p.w) + c_woff);
  f32x4 ra[DMA ? 1 : PF][DMA ? 1 : NA], rb[DMA ? 1 : PF][DMA ? 1 : NBV];  // 16-byte containers (4 floats or 8 halves)
  int kg = 0;
  auto gload_b = [&](int slot) {
#pragma unroll
    for (int j = 0; j < NBV; ++j) rb[slot][j] = dc_bload4(wr_, bvoff[j], (unsigned)kg * ES);
    kg += BK;
  };
  // DMA: the filter rows of one K tile into LDS stage `stg` (piece j = rows j*RPP + 8*wave .. +7 of the B block)
  auto dma_b = [&](int stg) {
#pragma unroll
    for (int j = 0; j < NBV; ++j) dc_dma16(wrs, ldsw + stg * TILEB + BM * LDB + j * RPP * LDB, bvoff[j], (unsigned)kg * ES);  // prologue: stage index
    kg += BK;
  };
  if constexpr (DMA) {
#pragma unroll
    for (int k = 0; k < DMA - 1; ++k)
      if (k < T_) dma_b(k);
  } else {
#pragma unroll
    for (int k = 0; k < PF; ++k)
      if (k < T_) gload_b(k);
  }

  // (2) epilogue constants (folded BatchNorm/Scale/bias) also travel now
  float sc[FN], sh[FN];
  float epi_c = 0.f;  // SWP: thread t < BN holds scale[n0 + t], BN <= t < 2*BN holds shift[n0 + t - BN]; to LDS before the first barrier
  if constexpr (SWP) {
    static_assert(2 * BN <= NT, "one epilogue constant per thread");
    const bool is_sh = t >= BN;
    const int co = n0 + (is_sh ? t - BN : t);
    epi_c = is_sh ? 0.f : 1.f;
    const i32x4 scr = dc_rsrc_words(p.scale), shr = dc_rsrc_words(p.shift);  // uniform descriptors (SGPRs)
    if (!is_sh) {
      if (p.scale && co < p.Cout) epi_c = dc_load_f32_untracked(scr, (unsigned)co * 4u);
    } else if (t < 2 * BN) {
      if (p.shift && co < p.Cout) epi_c = dc_load_f32_untracked(shr, (unsigned)co * 4u);
    }
  } else {
#pragma unroll
    for (int b = 0; b < FN; ++b) {
      const int co = n0 + wc * TN + b * 32 + (lane & 31);
      const bool cok = co < p.Cout;
      sc[b] = (cok && p.scale) ? p.scale[co] : 1.f;
      sh[b] = (cok && p.shift) ? p.shift[co] : 0.f;
    }
  }

  stamp(1);
  // (3) activation rows: every loading thread decodes ITS rows itself (two magic-number divisions and a handful of
  //     multiply-adds each) and requests their first tiles at once.  An earlier form decoded each row once, in one thread,
  //     and shared it through LDS: that put a barrier and an LDS round trip between the kernel arguments and the first
  //     activation load — ~2 k cycles of every workgroup's life.  Loop-invariant voffset + validity bit per tap (zero padding
  //     = out-of-range voffset); bit (ty*ntx + tx) = source row ty inside the image and source element inside its row (the
  //     thread's element column is added: it matters only for the stem's row taps, which span several pixels).
  unsigned avoff[NA], amask[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int m = m0 + lrow + RPP * i;
    avoff[i] = kOOB;
    amask[i] = 0;
    if (!MC && c_densex) {  // 1x1 over a dense tensor: pixel m starts at element m * sx, its one tap is always inside
      if (m < c_M) avoff[i] = (unsigned)(m * p.sx) * ES + lcb, amask[i] = 1u;
    } else if (m < c_M) {
      const int n = dc_fastdiv(m, c_dohw);
      const int rem = m - n * (c_OH * c_OW);
      const int oy = dc_fastdiv(rem, c_dow);
      const int ox = rem - oy * c_OW;
      avoff[i] = (unsigned)(int)(((long)n * c_ximg + (long)(oy * p.sy) * c_xrow + ox * p.sx) * ES) + lcb;
      unsigned rowmask = 0, colmask = 0, full = 0;
      if (DMA && c_nty == 3 && c_ntx == 3) {
        // the 3x3 layers (a third of the float16 time): the same masks, straight-line — the generic loops below cost ~3 k
        // cycles of every workgroup's prologue (dependent compares + branches, two to four rows per thread)
        const int y0 = oy * p.sy + c_dy0, x0 = ox * p.sx + lce + c_x0;
        const unsigned r0 = (unsigned)y0 < (unsigned)c_xrows, r1 = (unsigned)(y0 + c_ddy) < (unsigned)c_xrows, r2 = (unsigned)(y0 + 2 * c_ddy) < (unsigned)c_xrows;
        const unsigned cm = ((unsigned)x0 < (unsigned)c_xrowlen ? 1u : 0u) | ((unsigned)(x0 + c_ddx) < (unsigned)c_xrowlen ? 2u : 0u) |
                            ((unsigned)(x0 + 2 * c_ddx) < (unsigned)c_xrowlen ? 4u : 0u);
        amask[i] = (r0 ? cm : 0u) | (r1 ? cm << 3 : 0u) | (r2 ? cm << 6 : 0u);
        continue;
      }
#pragma nounroll
      for (int ty = 0; ty < c_nty; ++ty) rowmask |= ((unsigned)(oy * p.sy + c_dy0 + ty * c_ddy) < (unsigned)c_xrows ? 1u : 0u) << ty;
#pragma nounroll
      for (int tx = 0; tx < c_ntx; ++tx) colmask |= ((unsigned)(ox * p.sx + lce + c_x0 + tx * c_ddx) < (unsigned)c_xrowlen ? 1u : 0u) << tx;
#pragma nounroll
      for (int ty = 0; ty < c_nty; ++ty)
        if ((rowmask >> ty) & 1u) full |= colmask << (ty * c_ntx);
      amask[i] = full;
    }
  }
  // the source descriptor starts `x_bias` elements BEFORE the tensor so that every tap displacement is a
  // non-negative soffset; masked lanes never touch memory, valid lanes land inside the tensor
  const __amdgpu_buffer_rsrc_t xr = dc_rsrc(px + c_xbias, 0x7fffffffu);
  const i32x4 xrs = dc_rsrc_words(px + c_xbias);
  // tap cursor, all uniform (SALU): (tx, c0, running bit) and the element displacement of the current tap
  int tx = 0, c0 = 0, tbit = 0;
  int row_soff = c_dy0 * c_xrow + c_x0 - c_xbias;  // displacement of tap (ty, 0)
  int tap_soff = row_soff;
  auto gload_a = [&](int slot) {
    const unsigned soff = (unsigned)(tap_soff + c0) * ES;
    const unsigned bit = 1u << tbit;
#pragma unroll
    for (int i = 0; i < NA; ++i) ra[slot][i] = dc_bload4(xr, (amask[i] & bit) ? avoff[i] : kOOB, soff);
    c0 += BK;
    if (c0 >= p.klen) {
      c0 = 0;
      ++tbit;
      ++tx;
      tap_soff += c_ddx;
      if (tx >= c_ntx) {
        tx = 0;
        row_soff += c_ddy * c_xrow;
        tap_soff = row_soff;
      }
    }
  };
  // DMA: the tap cursor hands out (soffset, validity bit) of the next K tile ...
  unsigned d_soff = 0, d_bit = 0, d_kg = 0;
  auto dma_next = [&]() {
    d_soff = (unsigned)(tap_soff + c0) * ES;
    d_bit = 1u << tbit;
    d_kg = (unsigned)kg * ES;
    kg += BK;
    c0 += BK;
    if (c0 >= p.klen) {
      c0 = 0;
      ++tbit;
      ++tx;
      tap_soff += c_ddx;
      if (tx >= c_ntx) {
        tx = 0;
        row_soff += c_ddy * c_xrow;
        tap_soff = row_soff;
      }
    }
  };
  // ... and piece pc of that tile (0..NA-1: activation rows, NA..NA+NBV-1: filter rows) goes to stage `stg`
  auto dma_piece = [&](unsigned so, int pc) {  // so: byte offset of the stage
    if (pc < NA) {
#ifdef DC_ABL  // diagnostic builds (wrong results, timing only): bit 0 = no in-loop DMA, 1 = no in-loop barrier, 2 = no in-loop
               // fragment reads, 4 = activation pieces of 3 of 9 taps only (the bytes an LDS-resident halo would move)
      if ((DC_ABL & 16) && c_nty == 3 && (d_bit & 0x1b6u)) return;
#endif
      dc_dma16(xrs, ldsw + so + pc * RPP * LDB, (amask[pc < NA ? pc : 0] & d_bit) ? avoff[pc < NA ? pc : 0] : kOOB, d_soff);
    }
    else
      dc_dma16(wrs, ldsw + so + BM * LDB + (pc - NA) * RPP * LDB, bvoff[pc >= NA ? pc - NA : 0], d_kg);
  };
  if constexpr (DMA) {
    // the filter pieces of the first DMA-1 tiles are on their way already (dma_b advanced kg): only the activation pieces here
    const int kg_keep = kg;
#pragma unroll
    for (int k = 0; k < DMA - 1; ++k)
      if (k < T_) {
        dma_next();
#pragma unroll
        for (int i = 0; i < NA; ++i) dma_piece(k * TILEB, i);
      }
    kg = kg_keep;
    if (LATE && T_ >= DMA) {  // tile DMA-1, whole (activation and filter pieces), into the last stage
      dma_next();
#pragma unroll
      for (int pc = 0; pc < NA + NBV; ++pc) dma_piece((DMA - 1) * TILEB, pc);
    }
  } else {
#pragma unroll
    for (int k = 0; k < PF; ++k)
      if (k < T_) gload_a(k);
  }
  stamp(2);
  // (4) output byte offset of every tile row, for the epilogue (off the critical path: the loads are in flight)
  if (t < BM) {
    const int m = m0 + t;
    int yo = -1;
    if (!MC && c_densey) {
      if (m < c_M) yo = m * c_ypix * ES;
    } else if (m < c_M) {
      const int n = dc_fastdiv(m, c_dohw);
      const int rem = m - n * (c_OH * c_OW);
      const int oy = dc_fastdiv(rem, c_dow);
      const int ox = rem - oy * c_OW;
      yo = (int)(((long)n * c_yimg + (long)oy * c_yrow + (long)ox * c_ypix) * ES);
    }
    rowinfo[t] = yo;
  }
  stamp(3);
  const __amdgpu_buffer_rsrc_t yr = dc_rsrc(reinterpret_cast<T*>(c_y) + c_yoff, 0x7fffffffu);
  const __amdgpu_buffer_rsrc_t rr = dc_rsrc(reinterpret_cast<const T*>(c_resid ? c_resid : c_y) + c_yoff, 0x7fffffffu);

  f32x16 acc[FM][FN];
#pragma unroll
  for (int a = 0; a < FM; ++a)
#pragma unroll
    for (int b = 0; b < FN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  auto lstore = [&](int buf, int slot) {
    unsigned char* As = smem + buf * TILEB;
    unsigned char* Bs = As + BM * LDB;
#pragma unroll
    for (int i = 0; i < NA; ++i)
      *reinterpret_cast<f32x4*>(As + (lrow + RPP * i) * LDB + lcb) = ra[slot][i];
#pragma unroll
    for (int j = 0; j < NBV; ++j)
      *reinterpret_cast<f32x4*>(Bs + (lrow + RPP * j) * LDB + lcb) = rb[slot][j];
  };
  // MFMA operand fragments, two register sets (chunk parity); lane holds 16 bytes of row (lane&31):
  // fp32: k = 4*(lane>>5) .. +3 of the 8-deep chunk; fp16: k = 8*(lane>>5) .. +7 of the 16-deep chunk
  f32x4 av[2][FM], bv[2][FN];
  // DMA image: 16-byte chunk c of row r sits at position c ^ ((r >> 1) & 7); the wave's q-th operand chunk pair is chunk
  // 2*(q*WK + wk) + (lane >> 5), and since q*WK*32 only touches address bits 5..6 it is applied as an XOR on the offset
  const int frag_k = DMA ? (((2 * wk + (lane >> 5)) ^ (ROW256 ? (lane & 15) : ((lane >> 1) & 7))) * 16) : ((lane >> 5) * 16 + wk * 32);
  const int frag_a = (wr * TM + (lane & 31)) * LDB + frag_k;
  const int frag_b = BM * LDB + (wc * TN + (lane & 31)) * LDB + frag_k;
  auto frag_load = [&](int buf, int q, int set) {
    const unsigned char* As = smem + buf * TILEB + (DMA ? (frag_a ^ (q * WK * 32)) : frag_a + q * WK * 32);
    const unsigned char* Bs = smem + buf * TILEB + (DMA ? (frag_b ^ (q * WK * 32)) : frag_b + q * WK * 32);
#pragma unroll
    for (int a = 0; a < FM; ++a) av[set][a] = *reinterpret_cast<const f32x4*>(As + a * 32 * LDB);
#pragma unroll
    for (int b = 0; b < FN; ++b) bv[set][b] = *reinterpret_cast<const f32x4*>(Bs + b * 32 * LDB);
  };

  // ---- K loop: software pipeline -------------------------------------------------------------------
  // A wave issues in order, and on gfx950 the fp32 MFMA shares the SIMD's fp32 datapath with VALU, so
  // (a) VALU work in the loop is paid on top of the MFMAs -> there is almost none (buffer addressing);
  // (b) anything not interleaved BETWEEN MFMAs in program order stalls the matrix pipe -> the loop's
  //     non-MFMA work is placed into specific MFMA steps of the tile and pinned with sched_barrier.
  // Per tile a wave owns NCH 32-byte chunks = NST MFMA steps (4 per chunk in fp32, 1 in fp16):
  //   first step of chunk q: ds_read the fragments of chunk q+1
  //   step 1: ds_write tile it+1 (ring slot loaded PF iterations ago) into the idle LDS stage
  //   step 2/3: issue the global loads of tile it+1+PF into the ring slot just freed
  //   step GBAR: barrier, then ds_read chunk 0 of tile it+1 from the stage just filled — in fp32 its latency
  //            is covered by the 3 remaining MFMA steps of this tile (all fragment reads of the current
  //            stage are issued before the barrier, so the next tile may overwrite it).
  // Tile k lives in ring slot k % PF: filters are streamed from HBM once per forward (263 MB per image
  // sweep the 256 MB Infinity Cache), so a single tile of lookahead does not cover their latency.
  if constexpr (DMA) {
    // tile 0 has landed when at most the later prologue tiles' activation pieces (requested after it) are outstanding
    // (the early tiles' activation pieces after tile 0's, NA each, and the late tile's NA + NBV)
    const int pt = T_ < DMA - 1 ? T_ : DMA - 1;  // early tiles requested
    if (LATE && T_ >= DMA) {
      if (pt >= 3) dc_wait_vm<2 * NA + NA + NBV>();
      else if (pt == 2) dc_wait_vm<NA + NA + NBV>();
      else dc_wait_vm<NA + NBV>();
    } else {
      if (pt >= 3) dc_wait_vm<2 * NA>();
      else if (pt == 2) dc_wait_vm<NA>();
      else dc_wait_vm<0>();
    }
    if constexpr (SWP) {  // the constants were requested before every DMA piece: the wait above covers them
      if (t < 2 * BN) epi_sc[t] = epi_c;
    }
    dc_lds_barrier();  // every wave's pieces of tile 0 are in LDS; the rows' output offsets (and constants) are visible
  } else {
    lstore(0, 0);
    if (PF < T_) {
      gload_a(0);
      gload_b(0);
    }
    __syncthreads();  // tile 0 staged; the rows' output offsets are visible
  }
  frag_load(0, 0, 0);
  // (5) small wave tiles: the shortcut itself is requested now, before the K loop.
  //     MFMA 32x32 C layout: col = lane&31 (channel), row = (r&3)+8*(r>>2)+4*(lane>>5).
  //     With in-workgroup split-K every one of the WK waves finalises RPW of the 16 accumulator registers.
  float rs[EARLY_RESID ? FM * FN * RPW : 1];
  if (EARLY_RESID && c_resid) {
#pragma unroll
    for (int a = 0; a < FM; ++a)
#pragma unroll
      for (int b = 0; b < FN; ++b)
#pragma unroll
        for (int e = 0; e < RPW; ++e) {
          const int r = wk * RPW + e;
          const int yo = rowinfo[wr * TM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)];
          const int co = n0 + wc * TN + b * 32 + (lane & 31);
          const unsigned off = (yo >= 0 && co < p.Cout) ? (unsigned)yo + co * ES : kOOB;
          rs[(a * FN + b) * RPW + e] = Elem<T>::load(rr, off);
        }
  }
  stamp(4);
  // shortcut tile through LDS-DMA (swapped-operand epilogue, float16, one wave per output fragment set, a stage holds the tile)
  constexpr bool RD_OK = SWP && ES == 2 && WK == 1 && (BN * ES == 128 || BN * ES == 256) && BM * BN * ES <= TILEB && (BM * BN * ES / 1024) % NW == 0;
  constexpr int RD_PCS = RD_OK ? BM * BN * ES / 1024 / NW : 1;
  bool rd_on = false;
  unsigned rd_off = 0;
  if constexpr (DMA) {
    // ---- K loop, LDS-DMA form.  Tile `it` is computed from stage it % DMA while the pieces of tile it+DMA-1 are requested
    // into the stage tile it-1 was read from (free since the barrier taken inside tile it-1).  Before the last MFMA step of a
    // tile (fp32: the third last) every wave (a) waits until ITS pieces of tile it+1 have landed — vmcnt counts requests, the
    // younger tiles stay in flight —, (b) waits for its own fragment reads, (c) takes the one barrier of the tile: after it
    // tile it+1 is complete in LDS for every reader and stage it % DMA may be overwritten.  The first fragments of tile it+1
    // are fetched right there, under the remaining MFMA step(s).
    constexpr int PPW = NA + NBV;  // pieces (1 KiB wave requests) per wave and tile
    static_assert(2 * PPW <= 60, "vmcnt is a 6-bit counter");
    // The stage offsets are run-time scalars and the loop is NOT unrolled over the ring (one body for the steady state, one
    // for the last DMA-1 tiles, one for the last tile): with an unrolled ring and an early exit in the middle the
    // register allocator parked the 64 accumulators in different register classes on the two sides of the loop header
    // and copied all of them (v_accvgpr_read/write) once per round.
    rd_on = RD_OK && c_resid && p.vec_epi && !MC;
    unsigned sb = 0;                    // byte offset of the stage tile `it` is read from
    unsigned sbp = (DMA - 1) * TILEB;   // ... of the stage tile it+DMA-1 goes to (tile it-1 was read from it)
    auto tile = [&](auto more1_tag, auto moreD_tag, int it) {
      constexpr bool more1 = decltype(more1_tag)::value, moreD = decltype(moreD_tag)::value;
      const unsigned sbn = sb + TILEB == DMA * TILEB ? 0u : sb + TILEB;  // stage of tile it+1
      if (moreD) dma_next();
      // One MFMA step = FM*FN matrix instructions; its other work — the FM+FN fragment reads of the next chunk, the wait +
      // barrier + first reads of the next tile (step GBAR), this step's share of the DMA requests — is cut into FM*FN slices
      // and slice m is issued right AFTER matrix instruction m (pinned by sched_barrier): a wave issues in order and a
      // v_mfma_f32_32x32x16_f16 occupies the pipe for 32 cycles, so ~5 other instructions fit into each MFMA's shadow; issued
      // as one block before the step's MFMAs they left the matrix pipe idle for a third of a one-wave-per-SIMD K loop.
      constexpr int NM = FM * FN, NFR = FM + FN;
      auto frag_one = [&](unsigned so, int q, int set, int i) {  // i-th of the FM+FN 16-byte fragment reads of chunk q
        if (i < FM) av[set][i < FM ? i : 0] = *reinterpret_cast<const f32x4*>(smem + so + (frag_a ^ (q * WK * 32)) + (i < FM ? i : 0) * 32 * LDB);
        else bv[set][i >= FM ? i - FM : 0] = *reinterpret_cast<const f32x4*>(smem + so + (frag_b ^ (q * WK * 32)) + (i >= FM ? i - FM : 0) * 32 * LDB);
      };
#pragma unroll
      for (int g = 0; g < NST; ++g) {
        const int q = g / SPC, st = g % SPC, cur = q & 1;
        const bool rd_next = st == 0 && q + 1 < NCH;   // fragment reads of chunk q+1
        const bool sync = g == GBAR && more1;          // wait + barrier + first reads of tile it+1
        const int pc_lo = (g < GBAR && moreD) ? g * PPW / GBAR : 0, pc_hi = (g < GBAR && moreD) ? (g + 1) * PPW / GBAR : 0;
        const int n_sync = sync ? 1 + NFR : 0, n_rd = rd_next ? NFR : 0, n_items = n_sync + n_rd + (pc_hi - pc_lo);
#pragma unroll
        for (int m = 0; m < NM; ++m) {
          const int a = m / FN, b = m % FN;
          if constexpr (SPC == 4) {
            if constexpr (SWP) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[cur][b][st], av[cur][a][st], acc[a][b], 0, 0, 0);
            else acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][a][st], bv[cur][b][st], acc[a][b], 0, 0, 0);
          } else {
            if constexpr (SWP)
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bv[cur][b]),
                                                                 __builtin_bit_cast(f16x8, av[cur][a]), acc[a][b], 0, 0, 0);
            else
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av[cur][a]),
                                                                 __builtin_bit_cast(f16x8, bv[cur][b]), acc[a][b], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = m * n_items / NM; i < (m + 1) * n_items / NM; ++i) {
            if (i < n_sync) {
              if (i == 0) {
                // tiles that may stay in flight: those after it+1 that exist, but never a prologue tile (the prologue requested
                // all filter pieces before all activation pieces, so "the youngest n tiles" only means something for loop tiles)
                if (moreD && it >= DMA - 1) {
                  dc_wait_vm<(DMA - 2) * PPW>();  // steady state: the DMA-2 tiles after it+1 are loop tiles
                } else {
                  int lastq = it + DMA - 1;  // youngest tile requested so far (this tile's own requests precede this point)
                  if (lastq > T_ - 1) lastq = T_ - 1;
                  const int nearly = T_ < DMA - 1 ? T_ : DMA - 1;  // early prologue tiles 0..nearly-1: what is younger than the
                  int npro = (nearly - 1 < lastq ? nearly - 1 : lastq) - (it + 1);  // activation pieces of one of them is NA pieces per later one
                  if (npro < 0) npro = 0;
                  int nloop = lastq - (it + 1 > nearly - 1 ? it + 1 : nearly - 1);  // whole tiles (the late one, the loop's): PPW each
                  if (nloop < 0) nloop = 0;
                  if (DMA >= 4 && nloop == 2) dc_wait_vm<2 * PPW>();
                  else if (DMA >= 4 && nloop == 1 && npro == 1) dc_wait_vm<NA + PPW>();
                  else if (DMA >= 3 && nloop == 1) dc_wait_vm<PPW>();
                  else if (DMA >= 4 && npro == 2) dc_wait_vm<2 * NA>();
                  else if (DMA >= 3 && npro == 1) dc_wait_vm<NA>();
                  else dc_wait_vm<0>();
                }
#if defined(DC_ABL) && (DC_ABL & 2)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
                dc_lds_barrier();
#endif
              } else {
#if !(defined(DC_ABL) && (DC_ABL & 4))
                frag_one(sbn, 0, 0, i - 1);
#endif
              }
            } else if (i < n_sync + n_rd) {
#if !(defined(DC_ABL) && (DC_ABL & 4))
              frag_one(sb, q + 1, cur ^ 1, i - n_sync);
#endif
            } else {
#if !(defined(DC_ABL) && (DC_ABL & 1))
              dma_piece(sbp, pc_lo + (i - n_sync - n_rd));
#endif
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      sbp = sb;
      sb = sbn;
    };
    int it = 0;
    if (LATE && T_ >= DMA) {  // the prologue filled the whole ring: tile 0 has nothing to request
      tile(std::true_type{}, std::false_type{}, 0);
      it = 1;
    }
    for (; it < T_ - (DMA - 1); ++it) tile(std::true_type{}, std::true_type{}, it);
    for (; it < T_ - 1; ++it) tile(std::true_type{}, std::false_type{}, it);
    if constexpr (RD_OK) {
      // Shortcut tile by LDS-DMA: while the last K tile computes, the BM x BN shortcut values travel into the stage the
      // previous tile was read from (free since that tile's barrier) — whole 128- / 256-byte pixel rows per request (the
      // swapped-operand epilogue would otherwise fetch them as 32-byte fragments of 32 different lines per instruction),
      // no VGPRs, and a tile's time earlier than the epilogue could ask for them.  Same swizzle as the operand tiles.
      if (rd_on) {
        rd_off = sbp;
        constexpr int RB = BN * ES, LPR = RB / 16, PPP = 1024 / RB;  // row bytes, lanes per row, pixels per 1 KiB piece
        const i32x4 rrs = dc_rsrc_words(reinterpret_cast<const T*>(c_resid) + c_yoff);
#pragma unroll
        for (int j = 0; j < RD_PCS; ++j) {
          const int px_ = (j * NW + wave) * PPP + lane / LPR;        // tile row
          const int c = (lane % LPR) ^ (RB == 256 ? (px_ & 15) : ((px_ >> 1) & 7));
          const int yo = rowinfo[px_];
          const int co = n0 + c * VEC;
          const unsigned vo = (yo >= 0 && co < p.Cout) ? (unsigned)yo + (unsigned)co * ES : kOOB;
          dc_dma16(rrs, ldsw + sbp + j * NW * 1024, vo, 0);
        }
      }
    }
    tile(std::false_type{}, std::false_type{}, it);
  } else
  for (int it0 = 0; it0 < T_; it0 += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int it = it0 + u;
      if (it >= T_) break;
      const int buf = it & 1;
      const int slot = (u + 1) % PF;  // ring slot of tile it+1 (it0 is a multiple of PF)
      const bool more1 = it + 1 < T_, moreP = it + 1 + PF < T_;
#pragma unroll
      for (int g = 0; g < NST; ++g) {
        const int q = g / SPC, st = g % SPC, cur = q & 1;
        if (st == 0 && q + 1 < NCH) frag_load(buf, q + 1, cur ^ 1);
        if (g == GBAR && more1) {  // every fragment read of this stage has been issued (and is waited for here)
          __syncthreads();
          frag_load(buf ^ 1, 0, 0);
        }
        if (g == 1 && more1) lstore(buf ^ 1, slot);
        if (g == 2 && moreP) gload_a(slot);
        if (g == 3 && moreP) gload_b(slot);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int a = 0; a < FM; ++a)
#pragma unroll
          for (int b = 0; b < FN; ++b) {
            if constexpr (SPC == 4)
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][a][st], bv[cur][b][st], acc[a][b], 0, 0, 0);
            else
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av[cur][a]),
                                                                 __builtin_bit_cast(f16x8, bv[cur][b]), acc[a][b], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  stamp(5);
  // ---- epilogue ---------------------------------------------------------------------------------------
  // in-workgroup split-K: wave wk keeps registers [wk*RPW, (wk+1)*RPW) of every fragment and receives
  // the other waves' partials for them through LDS, so all waves store (no idle waves, 1/WK of the
  // LDS traffic of a gather-to-one reduction).  The body is instantiated once per value of wk (wave-uniform
  // branch) so that every accumulator index is a compile-time constant: no v_cndmask register selects.
  auto finish = [&](auto wk_tag) {
    constexpr int MYK = decltype(wk_tag)::value;
    const bool wide = WIDE_OK && p.wide_epi;  // uniform
    f32x4 wres[WIDE_OK ? WVEC : 1];             // shortcut vectors (8 halves each) of this thread's output vectors
    unsigned woff[WIDE_OK ? WVEC : 1];
    if constexpr (WIDE_OK) {
      if (wide) {
#pragma unroll
        for (int i = 0; i < WVEC; ++i) {
          const int v = t + i * NT, row = v / (BN / 8), cv = v - row * (BN / 8);
          const int yo = rowinfo[row];
          const int co = n0 + cv * 8;
          woff[i] = (yo >= 0 && co < p.Cout) ? (unsigned)yo + co * ES : kOOB;  // Cout % 8 == 0: a vector is all in or all out
          if (c_resid) wres[i] = dc_bload4(rr, woff[i], 0);
        }
      }
    }
    if (WK > 1) {
      __syncthreads();  // tile buffers are free
      float* part = reinterpret_cast<float*>(smem);
      // layout: [dst wave q][src wave (!= q) slot][wr*WC+wc][a][b][e][lane]
#pragma unroll
      for (int q = 0; q < WK; ++q) {
        if (q == MYK) continue;
        const int srcslot = MYK < q ? MYK : MYK - 1;
        float* dst = part + ((((q * (WK - 1) + srcslot) * WR * WC + wr * WC + wc) * FM * FN) * RPW) * 64 + lane;
#pragma unroll
        for (int a = 0; a < FM; ++a)
#pragma unroll
          for (int b = 0; b < FN; ++b)
#pragma unroll
            for (int e = 0; e < RPW; ++e) dst[((a * FN + b) * RPW + e) * 64] = acc[a][b][q * RPW + e];
      }
      __syncthreads();
#pragma unroll
      for (int sslot = 0; sslot < WK - 1; ++sslot) {
        const float* src = part + ((((MYK * (WK - 1) + sslot) * WR * WC + wr * WC + wc) * FM * FN) * RPW) * 64 + lane;
#pragma unroll
        for (int a = 0; a < FM; ++a)
#pragma unroll
          for (int b = 0; b < FN; ++b)
#pragma unroll
            for (int e = 0; e < RPW; ++e) acc[a][b][MYK * RPW + e] += src[((a * FN + b) * RPW + e) * 64];
      }
    }
    stamp(6);
    if constexpr (SWP) {
      // Swapped-operand accumulators: register r of fragment (a, b) is pixel wr*TM + a*32 + (lane & 31), channel
      // wc*TN + b*32 + 8*(r >> 2) + 4*(lane >> 5) + (r & 3): four runs of 4 consecutive channels.  float32: a run IS a 16-byte
      // vector.  float16: after the affine, v_permlane32_swap between the registers of runs g and g+1 leaves lanes 0..31 with
      // channels 8g..8g+7 and lanes 32..63 with 8g+8..8g+15 of their pixel: 16-byte vectors again, formed in registers.
      constexpr int NG = RPW / 4;  // runs per fragment this wave finalises
      const int h = lane >> 5;
      const float* scl = epi_sc;
      const float* shl = epi_sc + BN;
      int yo[FM];
#pragma unroll
      for (int a = 0; a < FM; ++a) yo[a] = rowinfo[wr * TM + a * 32 + (lane & 31)];
      if (p.vec_epi && (ES == 4 || NG >= 2)) {
        constexpr int NV = ES == 4 ? NG : (NG >= 2 ? NG / 2 : 1);  // 16-byte vectors per fragment
        f32x4 rv[FM][FN][NV];
        unsigned off[FM][FN][NV];
#pragma unroll
        for (int a = 0; a < FM; ++a)
#pragma unroll
          for (int b = 0; b < FN; ++b)
#pragma unroll
            for (int j = 0; j < NV; ++j) {
              const int g0 = MYK * NG + (ES == 4 ? j : 2 * j);
              const int co = n0 + wc * TN + b * 32 + (ES == 4 ? 8 * g0 + 4 * h : 8 * (g0 + h));
              off[a][b][j] = (yo[a] >= 0 && co < p.Cout) ? (unsigned)yo[a] + (unsigned)co * ES : kOOB;
              rv[a][b][j] = f32x4{0.f, 0.f, 0.f, 0.f};  // no shortcut: + 0 (one add instead of a select per element)
              if (c_resid && !rd_on) rv[a][b][j] = dc_bload4(rr, off[a][b][j], 0);
            }
        // ReLU is a template tag of `emit` below (uniform branch, like the shortcut): a ReLU-less layer stores its value
        // untouched, so a NaN / inf accumulator (a numerically broken model, a float16 overflow) propagates as it does in the
        // reference and in the element-wise path instead of being masked by max(x, -inf).
        if constexpr (RD_OK) {
          if (rd_on) {
            dc_wait_vm<0>();
            dc_lds_barrier();  // every wave's pieces of the shortcut tile have landed
            constexpr int RB = BN * ES;
#pragma unroll
            for (int a = 0; a < FM; ++a) {
              const int px_ = wr * TM + a * 32 + (lane & 31);
              const int sw = RB == 256 ? (px_ & 15) : ((px_ >> 1) & 7);
#pragma unroll
              for (int b = 0; b < FN; ++b)
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                  const int c = (wc * TN + b * 32 + 8 * (2 * j + h)) / 8;  // 16-byte chunk of the row (WK == 1: g0 = 2j)
                  rv[a][b][j] = *reinterpret_cast<const f32x4*>(smem + rd_off + px_ * RB + ((c ^ sw) * 16));
                }
            }
          }
        }
        // float16: the shortcut add reads its half straight from the packed register (v_fma_mix_f32: one instruction instead of a
        // conversion and an add — the same exactly rounded sum), two results are rounded and packed by one v_cvt_pk_f16_f32 and the
        // ReLU is a v_pk_max_f16 on the packed pair (rounding is monotonic and 0 / -inf are exact: max-then-round == round-then-max):
        // 28 VALU instructions per 16-byte vector with a shortcut, 20 without, instead of 48 — with three workgroups per CU
        // in their epilogues at once this phase is bound by VALU issue.
        auto emit = [&](auto res_tag, auto relu_tag) {
          constexpr bool HAS_RES = decltype(res_tag)::value, RELU = decltype(relu_tag)::value;
          const f16x2 rl2 = {(_Float16)0.f, (_Float16)0.f};
#pragma unroll
          for (int b = 0; b < FN; ++b)
#pragma unroll
            for (int j = 0; j < NV; ++j) {
              const int g0 = MYK * NG + (ES == 4 ? j : 2 * j);
              if constexpr (ES == 4) {
                const int cb = wc * TN + b * 32 + 8 * g0 + 4 * h;
                const f32x4 s4 = *reinterpret_cast<const f32x4*>(scl + cb), h4 = *reinterpret_cast<const f32x4*>(shl + cb);
#pragma unroll
                for (int a = 0; a < FM; ++a) {
                  f32x4 o;
#pragma unroll
                  for (int e = 0; e < 4; ++e) {
                    float v = acc[a][b][4 * g0 + e] * s4[e] + h4[e];
                    if (HAS_RES) v += rv[a][b][j][e];
                    o[e] = RELU ? fmaxf(v, 0.f) : v;
                  }
                  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), yr, off[a][b][j], 0, 0);
                }
              } else {
                const int cb0 = wc * TN + b * 32 + 8 * g0 + 4 * h, cb1 = cb0 + 8;
                const f32x4 s0 = *reinterpret_cast<const f32x4*>(scl + cb0), h0 = *reinterpret_cast<const f32x4*>(shl + cb0);
                const f32x4 s1 = *reinterpret_cast<const f32x4*>(scl + cb1), h1 = *reinterpret_cast<const f32x4*>(shl + cb1);
#pragma unroll
                for (int a = 0; a < FM; ++a) {
                  float lo[4], hi[4];
#pragma unroll
                  for (int e = 0; e < 4; ++e) {
                    lo[e] = acc[a][b][4 * g0 + e] * s0[e] + h0[e];
                    hi[e] = acc[a][b][4 * g0 + 4 + e] * s1[e] + h1[e];
                  }
                  dc_permlane32_swap4(lo, hi);
                  const u32x4 rz = __builtin_bit_cast(u32x4, rv[a][b][j]);
                  u32x4 o;
#pragma unroll
                  for (int i = 0; i < 4; ++i) {
                    float x0 = i < 2 ? lo[2 * i] : hi[2 * i - 4], x1 = i < 2 ? lo[2 * i + 1] : hi[2 * i - 3];
                    if (HAS_RES) {
                      x0 = dc_add_half_lo(rz[i], x0);
                      x1 = dc_add_half_hi(rz[i], x1);
                    }
                    const f32x2 xp = {x0, x1};
                    f16x2 hp = __builtin_convertvector(xp, f16x2);
                    if (RELU) hp = __builtin_elementwise_max(hp, rl2);
                    o[i] = __builtin_bit_cast(unsigned, hp);
                  }
                  __builtin_amdgcn_raw_buffer_store_b128(o, yr, off[a][b][j], 0, 0);
                }
              }
            }
        };
        if (c_resid) {
          if (p.relu) emit(std::true_type{}, std::true_type{});
          else emit(std::true_type{}, std::false_type{});
        } else {
          if (p.relu) emit(std::false_type{}, std::true_type{});
          else emit(std::false_type{}, std::false_type{});
        }
      } else {
        // element-wise form (odd channel counts, sigmoid heads, unaligned views, split-K 4 in float16)
#pragma unroll
        for (int b = 0; b < FN; ++b)
#pragma unroll
          for (int a = 0; a < FM; ++a)
#pragma unroll
            for (int gg = 0; gg < NG; ++gg) {
              const int g = MYK * NG + gg;
              const int cb = wc * TN + b * 32 + 8 * g + 4 * h;
              const f32x4 s4 = *reinterpret_cast<const f32x4*>(scl + cb), h4 = *reinterpret_cast<const f32x4*>(shl + cb);
              unsigned eo[4];
              float rvv[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const int co = n0 + cb + e;
                eo[e] = (yo[a] >= 0 && co < p.Cout) ? (unsigned)yo[a] + (unsigned)co * ES : kOOB;
                rvv[e] = c_resid ? Elem<T>::load(rr, eo[e]) : 0.f;
              }
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                float v = acc[a][b][4 * g + e] * s4[e] + h4[e] + rvv[e];
                if (p.relu) v = fmaxf(v, 0.f);
                if (n0 + cb + e < p.sigmoid_ch) v = 1.f / (1.f + expf(-v));
                Elem<T>::store(v, yr, eo[e]);
              }
            }
      }
      return;
    }
    if constexpr (WIDE_OK) {
      if (wide) {
        float* stg = reinterpret_cast<float*>(smem);
        __syncthreads();  // every wave is past its reads of the tile buffers / the split-K partials
#pragma unroll
        for (int b = 0; b < FN; ++b)
#pragma unroll
          for (int a = 0; a < FM; ++a)
#pragma unroll
            for (int e = 0; e < RPW; ++e) {
              const int r = MYK * RPW + e;
              const int row = wr * TM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
              stg[row * WPS + wc * TN + b * 32 + (lane & 31)] = acc[a][b][r] * sc[b] + sh[b];
            }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < WVEC; ++i) {
          const int v = t + i * NT, row = v / (BN / 8), cv = v - row * (BN / 8);
          const f32x4 lo = *reinterpret_cast<const f32x4*>(stg + row * WPS + cv * 8);
          const f32x4 hi = *reinterpret_cast<const f32x4*>(stg + row * WPS + cv * 8 + 4);
          f16x8 rz = __builtin_bit_cast(f16x8, wres[i]);
          f16x8 o;
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            float x = (q < 4 ? lo[q] : hi[q - 4]) + (c_resid ? (float)rz[q] : 0.f);
            if (p.relu) x = fmaxf(x, 0.f);
            o[q] = (_Float16)x;
          }
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), yr, woff[i], 0, 0);
        }
        return;
      }
    }
#pragma unroll
    for (int b = 0; b < FN; ++b) {
      const int co = n0 + wc * TN + b * 32 + (lane & 31);
      const bool cok = co < p.Cout;
      const bool sig = co < p.sigmoid_ch;
#pragma unroll
      for (int a = 0; a < FM; ++a) {
        unsigned off[RPW];
        float rv[RPW];
#pragma unroll
        for (int e = 0; e < RPW; ++e) {
          const int r = MYK * RPW + e;
          const int yo = rowinfo[wr * TM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)];
          off[e] = (yo >= 0 && cok) ? (unsigned)yo + co * ES : kOOB;  // masked lanes: load 0 / store dropped
        }
        if (EARLY_RESID) {
#pragma unroll
          for (int e = 0; e < RPW; ++e) rv[e] = c_resid ? rs[(a * FN + b) * RPW + e] : 0.f;
        } else if (c_resid) {  // all shortcut loads of the fragment in flight at once
#pragma unroll
          for (int e = 0; e < RPW; ++e) rv[e] = Elem<T>::load(rr, off[e]);
        } else {
#pragma unroll
          for (int e = 0; e < RPW; ++e) rv[e] = 0.f;
        }
#pragma unroll
        for (int e = 0; e < RPW; ++e) {
          float v = acc[a][b][MYK * RPW + e] * sc[b] + sh[b] + rv[e];
          if (p.relu) v = fmaxf(v, 0.f);
          if (sig) v = 1.f / (1.f + expf(-v));
          Elem<T>::store(v, yr, off[e]);
        }
      }
    }
  };
  if constexpr (WK == 1) {
    finish(std::integral_constant<int, 0>{});
  } else if constexpr (WK == 2) {
    if (wk == 0) finish(std::integral_constant<int, 0>{});
    else finish(std::integral_constant<int, 1>{});
  } else if constexpr (WK == 4) {
    if (wk == 0) finish(std::integral_constant<int, 0>{});
    else if (wk == 1) finish(std::integral_constant<int, 1>{});
    else if (wk == 2) finish(std::integral_constant<int, 2>{});
    else finish(std::integral_constant<int, 3>{});
  } else {
    static_assert(WK == 8, "WK is 1, 2, 4 or 8");
    if (wk == 0) finish(std::integral_constant<int, 0>{});
    else if (wk == 1) finish(std::integral_constant<int, 1>{});
    else if (wk == 2) finish(std::integral_constant<int, 2>{});
    else if (wk == 3) finish(std::integral_constant<int, 3>{});
    else if (wk == 4) finish(std::integral_constant<int, 4>{});
    else if (wk == 5) finish(std::integral_constant<int, 5>{});
    else if (wk == 6) finish(std::integral_constant<int, 6>{});
    else finish(std::integral_constant<int, 7>{});
  }
  stamp(7);
}

// magic {multiplier, shift word} for n / d, 0 <= n < 2^31 (see dc_fastdiv): sh = 31 + ceil(log2 d), mul = floor(2^sh / d) + 1
static void dc_magic(unsigned d, unsigned (&mg)[2]) {
  if (d <= 1) {
    mg[0] = 0;
    mg[1] = 0x80000000u;
    return;
  }
  int l = 0;
  while ((1ull << l) < d) ++l;
  const int sh = 31 + l;
  const unsigned long long q = (((unsigned __int128)1) << sh) / d;
  mg[0] = (unsigned)(q + 1);
  mg[1] = (unsigned)(sh - 32);
}

namespace {
struct VariantEntry {
  ConvVariant v;
  void (*kernel)(const ConvGemmParams);
  int BK;
  int esize;
  void (*kernel_mc)(const ConvGemmParams);  // multi-class instantiation (the deconvolution heads), or null
  void (*kernel_mp)(const ConvMultiArgs);   // multi-problem instantiation (pyramid-grouped launches)
};
// the three instantiations of one tile: single problem, multi-class (or null), multi-problem
#define DC_K3(T, BM, BN, BK, WR, WC, WK, PF, DMA, SWP, ES, WITH_MC)                                            \
  conv_gemm_kernel<T, BM, BN, BK, WR, WC, WK, PF, false, DMA, SWP>, BK, ES,                                     \
      WITH_MC ? conv_gemm_kernel<T, BM, BN, BK, WR, WC, WK, PF, WITH_MC, DMA, SWP> : nullptr,                   \
      conv_gemm_kernel<T, BM, BN, BK, WR, WC, WK, PF, false, DMA, SWP, true>
#define DC_VARIANT(BM, BN, BK, WR, WC, WK, PF) \
  { {#BM "x" #BN "x" #BK "_w" #WR #WC #WK "_p" #PF, BM, BN, WR, WC, WK}, DC_K3(float, BM, BN, BK, WR, WC, WK, PF, 0, false, 4, false) }
#define DC_VARIANT_MC(BM, BN, BK, WR, WC, WK, PF) \
  { {#BM "x" #BN "x" #BK "_w" #WR #WC #WK "_p" #PF, BM, BN, WR, WC, WK}, DC_K3(float, BM, BN, BK, WR, WC, WK, PF, 0, false, 4, true) }
#define DC_VARIANT_H(BM, BN, BK, WR, WC, WK, PF) \
  { {"h" #BM "x" #BN "x" #BK "_w" #WR #WC #WK "_p" #PF, BM, BN, WR, WC, WK}, DC_K3(_Float16, BM, BN, BK, WR, WC, WK, PF, 0, false, 2, false) }
#define DC_VARIANT_H_MC(BM, BN, BK, WR, WC, WK, PF) \
  { {"h" #BM "x" #BN "x" #BK "_w" #WR #WC #WK "_p" #PF, BM, BN, WR, WC, WK}, DC_K3(_Float16, BM, BN, BK, WR, WC, WK, PF, 0, false, 2, true) }
// LDS-DMA variants: "d" prefix, BK fixed by the 128-byte row (64 halves), S = LDS stages of the ring
#define DC_VARIANT_HD(BM, BN, WR, WC, WK, S) \
  { {"d" #BM "x" #BN "x64_w" #WR #WC #WK "_s" #S, BM, BN, WR, WC, WK}, DC_K3(_Float16, BM, BN, 64, WR, WC, WK, 1, S, true, 2, false) }
#define DC_VARIANT_HD_T(BM, BN, WR, WC, WK, S) /* LDS-transposed epilogue instead of the swapped-operand one (A/B) */ \
  { {"d" #BM "x" #BN "x64_w" #WR #WC #WK "_s" #S "_t", BM, BN, WR, WC, WK}, DC_K3(_Float16, BM, BN, 64, WR, WC, WK, 1, S, false, 2, false) }
#define DC_VARIANT_HD_MC(BM, BN, WR, WC, WK, S) \
  { {"d" #BM "x" #BN "x64_w" #WR #WC #WK "_s" #S, BM, BN, WR, WC, WK}, DC_K3(_Float16, BM, BN, 64, WR, WC, WK, 1, S, true, 2, true) }
#define DC_VARIANT_HD2(BM, BN, WR, WC, WK, S) /* 256-byte rows: BK = 128 halves */ \
  { {"d" #BM "x" #BN "x128_w" #WR #WC #WK "_s" #S, BM, BN, WR, WC, WK}, DC_K3(_Float16, BM, BN, 128, WR, WC, WK, 1, S, true, 2, false) }
#define DC_VARIANT_FD(BM, BN, BK, WR, WC, WK, S) /* float32: BK = 32 (128-byte rows) or 64 (256-byte rows) */ \
  { {"e" #BM "x" #BN "x" #BK "_w" #WR #WC #WK "_s" #S, BM, BN, WR, WC, WK}, DC_K3(float, BM, BN, BK, WR, WC, WK, 1, S, true, 4, false) }
const VariantEntry kVariants[] = {
    DC_VARIANT(128, 128, 32, 2, 2, 1, 2),  // 0: big-M layers (res2/res3)
    DC_VARIANT(128, 64, 32, 2, 2, 1, 2),   // 1
    DC_VARIANT(64, 128, 32, 2, 2, 1, 2),   // 2
    DC_VARIANT_MC(64, 64, 32, 2, 2, 1, 3), // 3
    DC_VARIANT_MC(64, 64, 64, 2, 2, 1, 3), // 4
    DC_VARIANT_MC(32, 64, 64, 1, 2, 2, 4), // 5: in-workgroup split-K 2
    DC_VARIANT(64, 32, 64, 2, 1, 2, 4),    // 6
    DC_VARIANT_MC(32, 32, 128, 1, 1, 4, 3), // 7: split-K 4 (tiny M*N, long K: res4/res5)
    DC_VARIANT_MC(32, 32, 64, 1, 1, 4, 4), // 8: same for K segments that are only multiples of 64
    DC_VARIANT(32, 64, 32, 1, 2, 2, 4),    // 9: K segments that are only multiples of 32 (the stem)
    // 8-wave workgroups: two waves per SIMD, so one wave's LDS/global/SALU work hides under the other's MFMAs
    DC_VARIANT_MC(32, 64, 64, 1, 2, 4, 4), // 10
    DC_VARIANT_MC(64, 64, 64, 2, 2, 2, 3), // 11
    DC_VARIANT_MC(128, 64, 32, 2, 2, 2, 2), // 12 (multi-class too: the merged heads fetch 8.7x their minimum on 32x32 tiles)
    DC_VARIANT_MC(128, 128, 32, 2, 2, 2, 2), // 13
    DC_VARIANT(64, 64, 32, 2, 2, 2, 3),    // 14
    DC_VARIANT_MC(32, 32, 128, 1, 1, 8, 3), // 15
    DC_VARIANT_MC(64, 128, 32, 2, 2, 2, 2), // 16
    // fp16 operands (v_mfma_f32_32x32x16_f16, fp32 accumulate); BK in halves: 64 = one 128-B line per row
    DC_VARIANT_H_MC(128, 128, 64, 2, 2, 1, 2),   // 17 (multi-class too: the float16 heads at batch 8 run 26 % faster on 128-wide tiles)
    DC_VARIANT_H_MC(128, 64, 64, 2, 2, 1, 2),    // 18
    DC_VARIANT_H(64, 128, 64, 2, 2, 1, 2),    // 19
    DC_VARIANT_H_MC(64, 64, 64, 2, 2, 1, 3),  // 20
    DC_VARIANT_H_MC(64, 64, 128, 2, 2, 2, 2), // 21: 8 waves
    DC_VARIANT_H_MC(32, 64, 128, 1, 2, 2, 3), // 22
    DC_VARIANT_H(64, 32, 128, 2, 1, 2, 3),    // 23
    DC_VARIANT_H_MC(32, 64, 256, 1, 2, 4, 2), // 24: 8 waves, split-K 4
    DC_VARIANT_H_MC(128, 128, 128, 2, 2, 2, 2),  // 25: 8 waves
    DC_VARIANT_H_MC(32, 32, 256, 1, 1, 4, 2), // 26
    // deeper rings for the short-K (bandwidth-class) layers: three of the four K tiles of a K = 256 layer are in flight at once
    DC_VARIANT_H(128, 128, 64, 2, 2, 1, 3),   // 27
    DC_VARIANT_H(128, 64, 64, 2, 2, 1, 3),    // 28
    DC_VARIANT_H(64, 128, 64, 2, 2, 1, 3),    // 29
    // 256-row / 256-column tiles (one workgroup per CU): half the operand traffic per flop for the long-K matrix-class layers
    DC_VARIANT_H(256, 128, 64, 4, 2, 1, 2),   // 30
    DC_VARIANT_H(128, 256, 64, 2, 4, 1, 2),   // 31
    // float16 through LDS-DMA (round 3): no register ring, no ds_write, unpadded swizzled stages, swapped-operand epilogue.
    // (Measured and dropped: 4 waves with a 3-stage ring on 128x128 — one workgroup per CU without a second wave per SIMD —,
    //  256-byte rows on 128x128, the LDS-transposed epilogue on the 8-wave tile.)
    DC_VARIANT_HD_MC(128, 128, 2, 2, 1, 2),   // 32: 66 KB -> two workgroups per CU
    DC_VARIANT_HD_MC(128, 128, 2, 2, 2, 3),   // 33: 8 waves, 98 KB, two tiles ahead: the 196-workgroup res4 layers
    DC_VARIANT_HD(128, 128, 2, 2, 2, 4),      // 34: 8 waves, 130 KB, three tiles ahead
    DC_VARIANT_HD(128, 64, 2, 2, 1, 3),       // 35: 73 KB -> two per CU
    DC_VARIANT_HD(64, 128, 2, 2, 1, 3),       // 36
    DC_VARIANT_HD(64, 64, 2, 2, 1, 4),        // 37: 65 KB
    DC_VARIANT_HD_MC(256, 128, 4, 2, 1, 3),   // 38: 146 KB, one per CU: the long-K matrix-class layers
    DC_VARIANT_HD_MC(128, 256, 2, 4, 1, 3),   // 39
    DC_VARIANT_HD(128, 64, 2, 2, 1, 2),       // 40: 49 KB -> three workgroups per CU (the bandwidth-class layers)
    DC_VARIANT_HD(64, 128, 2, 2, 1, 2),       // 41
    DC_VARIANT_HD(64, 64, 2, 2, 1, 2),        // 42: 33 KB
    DC_VARIANT_HD2(64, 64, 2, 2, 2, 2),       // 43: 256-byte rows, 8 waves
    DC_VARIANT_HD2(32, 64, 1, 2, 4, 3),       // 44: split-K 4 (small maps)
    DC_VARIANT_HD_T(128, 128, 2, 2, 1, 2),    // 45: LDS-transposed epilogue instead of the swapped-operand one
    DC_VARIANT_HD_T(64, 128, 2, 2, 1, 2),     // 46
    // float32 through LDS-DMA: the fp32 matrix pipe is 16x slower than the fp16 one, staging is not its limiter — these
    // tie with the register-ring tiles (+-3 % at batch 1, up to -5 % at batch 8: EXPERIMENTS.md A); the autotuner takes the wins
    DC_VARIANT_FD(32, 64, 64, 1, 2, 4, 3),    // 47: the res4/res5 batch-1 tile (8 waves, split-K 4)
    DC_VARIANT_FD(64, 128, 32, 2, 2, 2, 3),   // 48
    DC_VARIANT_FD(64, 64, 32, 2, 2, 2, 4),    // 49
    DC_VARIANT_FD(128, 64, 32, 2, 2, 2, 3),   // 50
    DC_VARIANT_FD(32, 32, 64, 1, 1, 4, 4),    // 51
    DC_VARIANT_FD(64, 64, 32, 2, 2, 1, 4),    // 52: 4 waves
    // 8 waves on 128x128 WITHOUT the in-workgroup split-K (wave tile 32x64 / 64x32): no exchange through LDS before the epilogue
    // (res4 3x3 21.5 -> 21.0 us, 1024->256 12.3 -> 11.7 at batch 8; bit-identical to the 4-wave tile's sums)
    DC_VARIANT_HD(128, 128, 4, 2, 1, 3),      // 53
    DC_VARIANT_HD(128, 128, 2, 4, 1, 3),      // 54
};
constexpr int kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);
}  // namespace

int conv_num_variants() { return kNumVariants; }
// (kStreamHalf — stream1x1.hip — answers these like a table entry: the group code handles it as one more multi-problem tile)
const ConvVariant& conv_variant(int i) {
  static const ConvVariant kStream = {"ws1x1", 32, 256, 1, 8, 1};
  return i == kStreamHalf ? kStream : kVariants[i].v;
}
int conv_variant_bk(int i) { return i == kStreamHalf ? 64 : kVariants[i].BK; }
int conv_variant_esize(int i) { return i == kStreamHalf ? 2 : kVariants[i].esize; }
bool conv_variant_multiclass(int i) { return i == kStreamHalf ? false : kVariants[i].kernel_mc != nullptr; }

long conv_grid(const ConvGemmParams& p, int variant) {
  const ConvVariant& v = kVariants[variant].v;
  const long tn = (p.Cout + v.BN - 1) / v.BN;
  if (p.ncls > 1) {
    long g = 0;
    for (int c = 0; c < p.ncls; ++c) g += (p.cls[c].M + v.BM - 1) / v.BM * tn;
    return g;
  }
  long tm = (p.M + v.BM - 1) / v.BM;
  return tm * tn;
}

int launch_conv_gemm(const ConvGemmParams& p_in, int variant, void* stream) {
  if (variant < 0 || variant >= kNumVariants) return (int)hipErrorInvalidValue;
  const VariantEntry& e = kVariants[variant];
  ConvGemmParams p = p_in;
  if (p.esize != e.esize) return (int)hipErrorInvalidValue;
  static const int wide_epi = getenv("DC_WIDE_EPI") ? atoi(getenv("DC_WIDE_EPI")) : 1;
  p.wide_epi = wide_epi && p.esize == 2 && p.ncls <= 1 && p.Cout % 8 == 0 && p.y_pix_stride % 8 == 0 && p.y_row_stride % 8 == 0 &&
               p.y_img_stride % 8 == 0 && p.sigmoid_ch == 0 && ((uintptr_t)p.y & 15) == 0 && (!p.resid || ((uintptr_t)p.resid & 15) == 0);
  {
    const long es = p.esize;
    p.vec_epi = p.ncls <= 1 && (p.Cout * es) % 16 == 0 && (p.y_pix_stride * es) % 16 == 0 && (p.y_row_stride * es) % 16 == 0 &&
                (p.y_img_stride * es) % 16 == 0 && p.sigmoid_ch == 0 && ((uintptr_t)p.y & 15) == 0 && (!p.resid || ((uintptr_t)p.resid & 15) == 0);
    static const int dense = getenv("DC_DENSE") ? atoi(getenv("DC_DENSE")) : 1;
    p.dense_x = dense && p.ncls <= 1 && p.nty == 1 && p.ntx == 1 && p.dy0 == 0 && p.x0 == 0 && p.sy == 1 && p.x_rows == p.OH &&
                p.x_row_stride == p.OW * p.sx && p.x_img_stride == (long)p.OH * p.x_row_stride && p.x_rowlen >= (p.OW - 1) * p.sx + p.klen;
    p.dense_y = dense && p.ncls <= 1 && p.y_row_stride == p.OW * p.y_pix_stride && p.y_img_stride == (long)p.OH * p.y_row_stride;
  }
  // n / d magic: sh = 31 + ceil(log2 d), mul = floor(2^sh / d) + 1, n/d = (n*mul) >> sh for 0 <= n < 2^31
  auto magic_of = [](unsigned d, unsigned (&mg)[2]) {
    if (d <= 1) {
      mg[0] = 0;
      mg[1] = 0x80000000u;
      return;
    }
    int l = 0;
    while ((1ull << l) < d) ++l;
    const int sh = 31 + l;
    const unsigned long long q = (((unsigned __int128)1) << sh) / d;
    mg[0] = (unsigned)(q + 1);
    mg[1] = (unsigned)(sh - 32);
  };
  if (p.ncls > 1) {
    // multi-class launch: the residue classes of a strided deconvolution as consecutive ranges of ONE grid
    if (p.ncls > kMaxClasses || !e.kernel_mc) return (int)hipErrorInvalidValue;
    const double lim = 2147483647.0;
    if ((double)e.esize * p.NB * (double)p.x_img_stride >= lim || (double)e.esize * p.NB * (double)p.y_img_stride >= lim)
      return (int)hipErrorInvalidValue;
    const long tn = (p.Cout + e.v.BN - 1) / e.v.BN;
    double wtot = 0, atot = 0;
    for (int c = 0; c < p.ncls; ++c) {
      ConvClass& q = p.cls[c];
      const int ntaps = q.nty * q.ntx;
      if (ntaps < 1 || ntaps > kMaxTaps || p.klen % e.BK != 0 || q.Ktot != ntaps * p.klen || q.M <= 0) return (int)hipErrorInvalidValue;
      int bias = 0;
      for (int ty : {0, q.nty - 1})
        for (int tx : {0, q.ntx - 1}) bias = std::min(bias, (q.dy0 + ty * q.ddy) * p.x_row_stride + q.x0 + tx * q.ddx);
      q.x_bias = bias;
      magic_of((unsigned)(q.OH * q.OW), q.div_ohw);
      magic_of((unsigned)q.OW, q.div_ow);
      q.tiles_m = (q.M + e.v.BM - 1) / e.v.BM;
      wtot += (double)e.esize * p.Cout * (double)q.Ktot;
      atot += (double)e.esize * q.M * (double)p.klen * q.nty;
    }
    if (wtot >= lim) return (int)hipErrorInvalidValue;
    p.tiles_n = (int)tn;
    magic_of((unsigned)tn, p.div_tn);
    p.xcd_on = 0;
    // XCD arrangement gx x gy minimising what one L2 has to fetch (its share of the filters + its share of the pixels)
    long blk = 0;
    {
      double best = 1e300;
      for (int lgx = 0; lgx <= 3; ++lgx) {
        const int gx = 1 << lgx, gy = 8 >> lgx;
        if (gx > tn) continue;
        long longest = 0;
        for (int xq = 0; xq < 8; ++xq) {
          const int qx = xq & (gx - 1), qy = xq >> lgx;
          const long ncnt = ((tn * (qx + 1)) >> lgx) - ((tn * qx) >> lgx);
          long cnt = 0;
          for (int c = 0; c < p.ncls; ++c) cnt += ((((long)p.cls[c].tiles_m * (qy + 1)) >> (3 - lgx)) - (((long)p.cls[c].tiles_m * qy) >> (3 - lgx))) * ncnt;
          longest = std::max(longest, cnt);
        }
        long total = 0;
        for (int c = 0; c < p.ncls; ++c) total += p.cls[c].tiles_m * tn;
        const double cost = (wtot / gx + atot / gy) * (1.0 + 0.02 * (longest * 8 - total) / (double)std::max(total, 1L));
        if (cost < best) best = cost, p.mc_lgx = lgx, blk = longest * 8;
      }
    }
    if (blk <= 0 || blk > 0x7fffffffL) return (int)hipErrorInvalidValue;
    const int nt = e.v.WR * e.v.WC * e.v.WK * 64;
    hipLaunchKernelGGL(e.kernel_mc, dim3((unsigned)blk), dim3(nt), 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
  }
  const int ntaps = p.nty * p.ntx;
  if (ntaps < 1 || ntaps > kMaxTaps || p.klen % e.BK != 0 || p.Ktot != ntaps * p.klen) return (int)hipErrorInvalidValue;
  // buffer (V#) addressing carries 32-bit byte offsets: every tensor of a launch must stay below 2 GiB
  {
    const double lim = 2147483647.0;
    const double xb = (double)e.esize * (double)p.NB * (double)p.x_img_stride;
    const double yb = (double)e.esize * ((double)p.NB * (double)p.y_img_stride);
    const double wb = (double)e.esize * (double)p.Cout * (double)p.Ktot;
    if (xb >= lim || yb >= lim || wb >= lim) return (int)hipErrorInvalidValue;
  }
  int bias = 0;  // most negative tap displacement: one of the four corners of the arithmetic grid
  for (int ty : {0, p.nty - 1})
    for (int tx : {0, p.ntx - 1})
      bias = std::min(bias, (p.dy0 + ty * p.ddy) * p.x_row_stride + p.x0 + tx * p.ddx);
  p.x_bias = bias;
  long grid = conv_grid(p, variant);
  if (grid <= 0) return 0;
  auto& magic = magic_of;
  const long tn = (p.Cout + e.v.BN - 1) / e.v.BN, tm = (p.M + e.v.BM - 1) / e.v.BM;
  p.tiles_n = (int)tn;
  magic((unsigned)tn, p.div_tn);
  magic((unsigned)(p.OH * p.OW), p.div_ohw);
  magic((unsigned)p.OW, p.div_ow);
  static const int xcd_map = getenv("DC_XCD_MAP") ? atoi(getenv("DC_XCD_MAP")) : 1;
  p.xcd_on = 0;
  p.tiles_m = (int)tm;
  if (xcd_map && grid >= 16) {
    // Blocks are observed to land on XCD (blockIdx % 8), each with its own 4 MB L2.  Cut the tile grid into 8
    // rectangles (gx along n, 8/gx along m) minimising the bytes an L2 must fetch for its rectangle
    // (filters of its n-range + pixels of its m-range); XCD q walks rectangle q.  A locality hint only.
    double best = 1e300;
    int best_lgx = -1;
    long best_grid = 0;
    for (int lgx = 0; lgx <= 3; ++lgx) {
      const int gx = 1 << lgx, gy = 8 >> lgx;
      if (gx > tn || gy > tm) continue;
      long maxrect = 0;
      for (int q = 0; q < 8; ++q) {
        const int qx = q & (gx - 1), qy = q >> lgx;
        const long rw = ((tn * (qx + 1)) >> lgx) - ((tn * qx) >> lgx), rh = ((tm * (qy + 1)) >> (3 - lgx)) - ((tm * qy) >> (3 - lgx));
        maxrect = std::max(maxrect, rw * rh);
      }
      const double w_bytes = (double)p.Cout * p.Ktot / gx, a_bytes = (double)p.M * p.klen * p.nty / gy;
      const double cost = (w_bytes + a_bytes) * (1.0 + 0.02 * (maxrect * 8 - grid) / (double)grid);
      if (cost < best) best = cost, best_lgx = lgx, best_grid = maxrect * 8;
    }
    if (best_lgx >= 0) {
      p.xcd_lgx = best_lgx;
      const unsigned w0 = (unsigned)(tn >> best_lgx);
      magic(std::max(w0, 1u), p.div_rw[0]);
      magic(w0 + 1, p.div_rw[1]);
      p.xcd_on = 1;
      grid = best_grid;
    }
  }
  const int nt = e.v.WR * e.v.WC * e.v.WK * 64;
  hipLaunchKernelGGL(e.kernel, dim3((unsigned)grid), dim3(nt), 0, (hipStream_t)stream, p);
  return (int)hipGetLastError();
}

bool conv_variant_multiproblem(int i) { return i == kStreamHalf ? true : kVariants[i].kernel_mp != nullptr; }

// Multi-problem launch: host-side preparation (once per plan), see kernels.h.
long prepare_conv_multi(ConvGemmParams& p, ConvMultiTable& tb, int nprob, int variant) {
  if (variant == kStreamHalf) {
    const long g = stream1x1_prepare_multi(p, tb, nprob);
    if (g > 0) p.nprob = nprob, p.ncls = 0;
    return g;
  }
  if (variant < 0 || variant >= kNumVariants || nprob < 1 || nprob > kMaxProblems) return -1;
  const VariantEntry& e = kVariants[variant];
  if (!e.kernel_mp || p.esize != e.esize || p.klen % e.BK != 0) return -1;
  const long es = p.esize;
  const double lim = 2147483647.0;
  static const int dense = getenv("DC_DENSE") ? atoi(getenv("DC_DENSE")) : 1;
  const long tn = (p.Cout + e.v.BN - 1) / e.v.BN;
  bool vec = (p.Cout * es) % 16 == 0 && p.sigmoid_ch == 0;
  double wtot = 0, atot = 0;
  long w_lo = 0, w_hi = 0;
  for (int c = 0; c < nprob; ++c) {
    ConvProblem& q = tb.prob[c];
    const int ntaps = q.nty * q.ntx;
    if (ntaps < 1 || ntaps > kMaxTaps || q.Ktot != ntaps * p.klen || q.M <= 0 || q.NB <= 0) return -1;
    if ((double)es * q.NB * (double)q.x_img_stride >= lim || (double)es * q.NB * (double)q.y_img_stride >= lim) return -1;
    if ((q.resid != nullptr) != (tb.prob[0].resid != nullptr)) return -1;
    int bias = 0;
    for (int ty : {0, q.nty - 1})
      for (int tx : {0, q.ntx - 1}) bias = std::min(bias, (q.dy0 + ty * q.ddy) * q.x_row_stride + q.x0 + tx * q.ddx);
    q.x_bias = bias;
    dc_magic((unsigned)(q.OH * q.OW), q.div_ohw);
    dc_magic((unsigned)q.OW, q.div_ow);
    q.tiles_m = (q.M + e.v.BM - 1) / e.v.BM;
    q.dense_x = dense && q.nty == 1 && q.ntx == 1 && q.dy0 == 0 && q.x0 == 0 && p.sy == 1 && q.x_rows == q.OH && q.x_row_stride == q.OW * p.sx &&
                q.x_img_stride == (long)q.OH * q.x_row_stride && q.x_rowlen >= (q.OW - 1) * p.sx + p.klen;
    q.dense_y = dense && q.y_row_stride == q.OW * q.y_pix_stride && q.y_img_stride == (long)q.OH * q.y_row_stride;
    vec = vec && (q.y_pix_stride * es) % 16 == 0 && (q.y_row_stride * es) % 16 == 0 && (q.y_img_stride * es) % 16 == 0 && ((uintptr_t)q.y & 15) == 0 &&
          (!q.resid || ((uintptr_t)q.resid & 15) == 0);
    w_lo = std::min(w_lo, q.w_off);
    w_hi = std::max(w_hi, q.w_off + (long)p.Cout * q.Ktot);
    atot += (double)es * q.M * (double)p.klen * q.nty;
  }
  wtot = (double)es * (double)(w_hi - w_lo);
  if (wtot >= lim) return -1;
  p.nprob = nprob;
  p.ncls = 0;
  p.vec_epi = vec ? 1 : 0;
  p.wide_epi = 0;  // (the LDS-transposed float16 epilogue addresses one tensor: the element-wise form takes its place here)
  {
    static const int wide_epi = getenv("DC_WIDE_EPI") ? atoi(getenv("DC_WIDE_EPI")) : 1;
    bool w8 = wide_epi && p.esize == 2 && p.Cout % 8 == 0 && p.sigmoid_ch == 0;
    for (int c = 0; c < nprob; ++c) {
      const ConvProblem& q = tb.prob[c];
      w8 = w8 && q.y_pix_stride % 8 == 0 && q.y_row_stride % 8 == 0 && q.y_img_stride % 8 == 0 && ((uintptr_t)q.y & 15) == 0 &&
           (!q.resid || ((uintptr_t)q.resid & 15) == 0);
    }
    p.wide_epi = w8 ? 1 : 0;
  }
  p.dense_x = p.dense_y = 0;
  p.xcd_on = 0;
  p.tiles_n = (int)tn;
  dc_magic((unsigned)tn, p.div_tn);
  // XCD arrangement gx x gy minimising what one L2 has to fetch (its share of the filters + its share of the pixels)
  long blk = 0;
  double best = 1e300;
  long total = 0;
  for (int c = 0; c < nprob; ++c) total += tb.prob[c].tiles_m * tn;
  for (int lgx = 0; lgx <= 3; ++lgx) {
    const int gx = 1 << lgx;
    if (gx > tn) continue;
    long longest = 0;
    for (int xq = 0; xq < 8; ++xq) {
      const int qx = xq & (gx - 1), qy = xq >> lgx;
      const long ncnt = ((tn * (qx + 1)) >> lgx) - ((tn * qx) >> lgx);
      long cnt = 0;
      for (int c = 0; c < nprob; ++c)
        cnt += ((((long)tb.prob[c].tiles_m * (qy + 1)) >> (3 - lgx)) - (((long)tb.prob[c].tiles_m * qy) >> (3 - lgx))) * ncnt;
      longest = std::max(longest, cnt);
    }
    const double cost = (wtot / gx + atot / (8 >> lgx)) * (1.0 + 0.02 * (longest * 8 - total) / (double)std::max(total, 1L));
    if (cost < best) best = cost, p.mc_lgx = lgx, blk = longest * 8;
  }
  if (blk <= 0 || blk > 0x7fffffffL) return -1;
  {
    const int lgx = p.mc_lgx, lgy = 3 - lgx;
    const unsigned w0 = (unsigned)(tn >> lgx);
    dc_magic(std::max(w0, 1u), p.div_rw[0]);
    dc_magic(w0 + 1, p.div_rw[1]);
    for (int qy = 0; qy < 8; ++qy) {
      int run = 0;
      for (int c = 0; c < kMaxProblems; ++c) {
        if (c < nprob && qy < (8 >> lgx)) {
          run += (int)((((long)tb.prob[c].tiles_m * (qy + 1)) >> lgy) - (((long)tb.prob[c].tiles_m * qy) >> lgy));
          tb.end[qy][c] = run;
        } else {
          tb.end[qy][c] = 0x7fffffff;
        }
      }
    }
  }
  return blk;
}

int launch_conv_multi(const ConvMultiArgs& a, int variant, long grid, void* stream) {
  if (variant == kStreamHalf) return launch_stream1x1_multi(a, stream);
  if (variant < 0 || variant >= kNumVariants || !kVariants[variant].kernel_mp || a.p.nprob < 1 || grid <= 0) return (int)hipErrorInvalidValue;
  const VariantEntry& e = kVariants[variant];
  const int nt = e.v.WR * e.v.WC * e.v.WK * 64;
  hipLaunchKernelGGL(e.kernel_mp, dim3((unsigned)grid), dim3(nt), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

// =====================================================================================================================
// Winograd F(2x2, 3x3): Y = A^T [ (G g G^T) . (B^T d B) ] A per 4x4 input patch d -> 2x2 outputs, summed over input
// channels as 16 independent GEMMs (one per transform position (i, j)): 2.25x fewer MFMA flops than the direct form.
//
// At batch 1 a res4 layer has only 391 tiles x 256 channels, so the kernel is built around operand TRAFFIC, not flops:
//  * workgroup = 4 x 8 tiles (two 16-tile MFMA fragments) x 16 output channels, 8 waves = (transform row i) x (fragment);
//    a wave owns the 4 positions (i, 0..3) of its fragment: 4 accumulators of v_mfma_f32_16x16x4_f32;
//  * the 10 x 18 input pixels the block reads are staged ONCE per 32 channels in LDS (ring of 3, one barrier per 32
//    channels); every wave reads the two patch rows its transform row needs and does B^T d B in registers (packed fp32)
//    right before its MFMAs — the transformed input never exists in memory;
//  * the transformed filters are the big stream (16/9 of the filter bytes, no reuse inside a workgroup): pre-packed on
//    the host so that each wave reads its B fragments straight from global memory, 1 KB contiguous per load, one
//    sub-step ahead; workgroups that share them (same 16 output channels) are adjacent in the grid;
//  * ~110 VGPRs and 79 KB of LDS: two workgroups per CU, so that forwards in flight can share CUs (with the register-
//    hungrier pipelined variant of the probe the kernel was as fast alone but worth nothing with three forwards in flight);
//  * the inverse transform reduces over j in registers and over i (four waves) through LDS, then applies the folded
//    BatchNorm/Scale affine, the shortcut and ReLU like the gather-GEMM's epilogue.
// Round 5 (tools/probes/winograd16_probe.hip, profiles/r05_winograd16_probe.txt):
//  * LDS row pitch 672 floats, no skew.  A ds_read_b128 is served in four groups of 16 lanes ({0-3,12-15,20-27}, {4-11,16-19,
//    28-31}, ...: MI355X_MICROARCH.md, LDS), one LDS cycle per group whose lanes touch 64 distinct banks.  A fragment read
//    addresses row(r) * pitch + 2 c * WPSTR + 4 kg floats (r = lane[3] tile row, c = lane[2:0] tile column, kg = lane[5:4]
//    channel quad); with WPSTR = 36 the groups are conflict-free iff two rows' pitch is a multiple of 64 floats: 2 * 672 =
//    21 * 64.  The round-1 layout (pitch 648 + a 4-float skew per row pair) paid 8 LDS cycles per read instead of 4;
//  * buffer (V#) addressing for the staged pixels and the filter fragments: loop-invariant per-thread voffsets, the channel
//    step in soffset, zero padding = an out-of-range voffset — no 64-bit address arithmetic, no predicated loads, no zero
//    fills in the K loop (70 -> 44 VALU instructions per 32 MFMAs: on gfx950 VALU work is paid on top of fp32 MFMA work);
//  * NG = 2 (tile name wino_f23_w16): SIXTEEN waves, the two 16-channel sub-steps of a staged step dealt to two groups of
//    eight.  A launch of at most one workgroup per CU (res4 at batch 1: 240) leaves two waves per SIMD, which cannot cover
//    each other's barrier, LDS and transform phases (K loop 27.3 k cycles for 16.4 k of MFMA); four waves per SIMD from ONE
//    workgroup do (23.3 k), once the LDS reads are conflict-free (with the old layout the 16-wave form was LDS-bound and
//    slower: 35.5 k).  Under load (grids of several rounds, forwards in flight) two 8-wave workgroups per CU are faster
//    than one 16-wave one: the autotuner decides per shape, tune_in_flight under the caller's load.
// Measured on the res4 3x3 shape (1x34x46, 256 -> 256; operands rotated through 355 MB): round-1 kernel 18.7 us alone /
// 13.4 us per image at 8 images per launch; NG = 1 now 17.5 / 11.7; NG = 2 15.95 / 13.5.
namespace {
constexpr int WBTY = 4, WBTX = 8, WBN = 16, WKC = 32;
constexpr int WRH = 2 * WBTY + 2, WRW = 2 * WBTX + 2;   // staged pixels: 10 x 18
constexpr int WPSTR = WKC + 4;                           // floats per staged pixel
constexpr int WPITCH = 672;                              // floats per staged pixel row: >= WRW * WPSTR = 648, and 2 * WPITCH % 64 == 0
constexpr int WSTAGE = WRH * WPITCH + 8;                 // + the dump slot of the staging threads past the block
constexpr int WNTH = 512;
static_assert(WPITCH >= WRW * WPSTR && (2 * WPITCH) % 64 == 0 && WPSTR == 36, "conflict-free ds_read_b128 layout (see above)");
static_assert(WPITCH % 4 == 0 && WPSTR % 4 == 0 && WSTAGE % 4 == 0, "16-byte units");
__device__ __forceinline__ int wino_rowbase(int row) { return row * WPITCH; }
__device__ __forceinline__ f32x2 wlo(f32x4 v) { return __builtin_shufflevector(v, v, 0, 1); }
__device__ __forceinline__ f32x2 whi(f32x4 v) { return __builtin_shufflevector(v, v, 2, 3); }
}  // namespace

template <int NG>
__global__ __launch_bounds__(WNTH * NG, 4) void wino_f23_kernel(const ConvGemmParams p) {
  constexpr int NTH = WNTH * NG;
  constexpr int WNLD = (WRH * WRW * (WKC / 4) + NTH - 1) / NTH;
  const long long t_entry = (long long)__builtin_amdgcn_s_memrealtime();
  DC_KARG_TOUCH(ka0, ka1, ka2, ka3, ka4);
  __shared__ __attribute__((aligned(16))) float stage[3][WSTAGE];
  // [g][i][b][tf][r][lane] partial inverse transforms: reuses the staging ring once the K loop is over (79 KB per
  // workgroup: two 8-wave workgroups fit the 160 KB of a CU)
  float (*part)[4][2][2][4][64] = reinterpret_cast<float (*)[4][2][2][4][64]>(&stage[0][0]);
  static_assert(sizeof(float) * NG * 4 * 2 * 2 * 4 * 64 <= sizeof(stage), "partials must fit in the staging ring");
  static_assert(2 * sizeof(stage) <= 160 * 1024, "two workgroups per CU");
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  auto stamp = [&](int slot) {  // DC_DEBUG_TIMING: per-wave phase stamps (see conv_gemm_kernel)
    if (p.dbg && lane == 0) {
      long long* d = p.dbg + ((long)blockIdx.x * (8 * NG) + wave) * 12;
      d[slot] = (long long)__builtin_readcyclecounter();
      if (slot == 0) d[8] = t_entry, d[10] = (long long)__builtin_amdgcn_s_memrealtime();
      if (slot == 7) d[9] = (long long)__builtin_amdgcn_s_memrealtime();
    }
  };
  stamp(0);
  const int C = p.klen, H = p.x_rows, W = p.x_rowlen / p.klen;
  // dilation d: the image is d*d interleaved phase images, each an ordinary pad-1 3x3 problem on the pixels
  // (phy + d*u, phx + d*v); tiles, blocks and staged coordinates below live on the phase grid (u, v)
  // (the tile grid and the magic numbers of the block-index divisions come from the host: seven runtime integer
  // divisions were 2.8 k cycles of every workgroup's life)
  const int d = p.ddy;
  const int NBY = p.w_NBY, NBX = p.w_NBX;
  const int nblk = p.w_nblk;
  // Workgroup b runs on XCD (b % 8), each XCD with its own L2.  The transformed filters are the big stream (16/9 of the
  // filter bytes), shared by the nblk workgroups of a 16-channel slice: hand every XCD a CONTIGUOUS range of the
  // (slice-major) logical grid, so that a slice is fetched from HBM by one L2 (two at a range boundary) instead of by
  // all eight.  A locality hint only: any bijection of the grid computes the same result.
  int lb = blockIdx.x;
  if (p.xcd_on) {
    const int g8 = gridDim.x >> 3, r8 = gridDim.x & 7, q = blockIdx.x & 7;
    lb = q * g8 + min(q, r8) + (blockIdx.x >> 3);
  }
  const int nt = dc_fastdiv(lb, p.w_div_nblk), blk = lb - nt * nblk;  // same-filter workgroups are adjacent in the logical grid
  const int nph = dc_fastdiv(blk, p.w_div_nbyx), brem = blk - nph * (NBY * NBX);
  const int n = dc_fastdiv(nph, p.w_div_dd), ph = nph - n * (d * d);
  const int phy = dc_fastdiv(ph, p.w_div_d), phx = ph - phy * d;
  const int by = dc_fastdiv(brem, p.w_div_nbx), bx = brem - by * NBX;
  const int oy0 = 2 * WBTY * by - 1, ox0 = 2 * WBTX * bx - 1;  // phase-grid coordinates of staged pixel (0, 0): pad 1
  DC_KARG_HOLD(ka0, ka1, ka2, ka3, ka4);  // the block decode above needed kernel arguments: the dummy loads have landed
  const int kg = lane >> 4;
  const int grp = NG == 1 ? 0 : __builtin_amdgcn_readfirstlane(wave >> 3);  // (an SGPR: it enters the filter loads' soffset)
  const int i = wave & 3, tf = (wave >> 2) & 1;
  // B^T row i as a combination of two patch rows: i=0: d0-d2, 1: d1+d2, 2: d2-d1, 3: d1-d3
  const int ra = i == 0 ? 0 : (i == 2 ? 2 : 1), rb = i == 0 ? 2 : (i == 1 ? 2 : (i == 2 ? 1 : 3));
  const float sb = i == 1 ? 1.f : -1.f;
  // V# addressing (see dc_rsrc): per-thread byte offsets are loop invariant, the 32-channel step travels in soffset, a pixel
  // outside the image is an out-of-range voffset (zeros); the image base (n is uniform) sits in the descriptor
  const __amdgpu_buffer_rsrc_t xr = dc_rsrc(reinterpret_cast<const float*>(p.x) + (long)n * p.x_img_stride, 0x7fffffffu);
  const __amdgpu_buffer_rsrc_t ur = dc_rsrc(p.w, 0x7fffffffu);
  unsigned gofs[WNLD];
  int sofs[WNLD];
#pragma unroll
  for (int q = 0; q < WNLD; ++q) {
    const int e = t + q * NTH;
    const int pix = e / (WKC / 4), cq = e % (WKC / 4);
    const int py = pix / WRW, px = pix % WRW;
    const int iy = phy + d * (oy0 + py), ix = phx + d * (ox0 + px);
    const bool ok = pix < WRH * WRW && oy0 + py >= 0 && ox0 + px >= 0 && iy < H && ix < W;
    gofs[q] = ok ? (unsigned)(iy * p.x_row_stride + ix * C + cq * 4) * 4u : kOOB;
    sofs[q] = (pix < WRH * WRW ? wino_rowbase(py) + px * WPSTR + cq * 4 : WSTAGE - 8 + (t & 1) * 4) >> 2;  // in float4 units (past the block: the dump slot)
  }
  const int r = (lane & 15) >> 3, c = lane & 7;
  const int ofs_a = wino_rowbase(2 * (2 * tf + r) + ra) + 2 * c * WPSTR + kg * 4;
  const int ofs_b = wino_rowbase(2 * (2 * tf + r) + rb) + 2 * c * WPSTR + kg * 4;
  f32x4 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned uvo = ((unsigned)((nt * 4 + i) * (C / 16)) * (4 * 64 * 4) + (unsigned)lane * 4u) * 4u;  // bytes (< 2 GiB: wino_eligible)
  // Registers are kept to 110 per wave on purpose: four waves per SIMD = two workgroups per CU (this kernel's, or one of
  // the gather-GEMM's), which is what lets forwards in flight share a CU; LDS reads are therefore issued right before
  // their use (the other resident waves hide their latency) and only the filter fragments run one sub-step ahead.
  f32x4 g[WNLD], b[2][4], da[4], db[4];
  auto gload = [&](int K) {
#pragma unroll
    for (int q = 0; q < WNLD; ++q) g[q] = dc_bload4(xr, gofs[q], (unsigned)(K * WKC * 4));
  };
  // (the stage is indexed in 16-byte units: the compiler cannot prove the alignment of a float index and would split every store
  // into two ds_write2_b32, whose lanes — 16 bytes apart — collide four ways on the 32 write banks)
  auto sstore = [&](int buf) {
#pragma unroll
    for (int q = 0; q < WNLD; ++q) reinterpret_cast<f32x4*>(&stage[buf][0])[sofs[q]] = g[q];
  };
  auto bload = [&](int slot, int k16) {
#pragma unroll
    for (int j = 0; j < 4; ++j) b[slot][j] = dc_bload4(ur, uvo + (unsigned)j * 1024u, (unsigned)k16 * 4096u);
  };
  auto lread = [&](int buf, int h) {
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
      da[c4] = *reinterpret_cast<const f32x4*>(&stage[buf][ofs_a + c4 * WPSTR + h * 16]);
      db[c4] = *reinterpret_cast<const f32x4*>(&stage[buf][ofs_b + c4 * WPSTR + h * 16]);
    }
  };
  auto compute = [&](int bslot) {
    f32x2 tl[4], th[4];
    const f32x2 sb2 = {sb, sb};
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
      tl[c4] = wlo(da[c4]) + sb2 * wlo(db[c4]);
      th[c4] = whi(da[c4]) + sb2 * whi(db[c4]);
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
  const int NS = C / WKC;
  // the epilogue's per-channel constants are requested here (two registers carried through the K loop) instead of behind it,
  // where their round trip was exposed
  const int co = nt * WBN + (lane & 15);
  const float sc = p.scale ? p.scale[co] : 1.f, sh = p.shift ? p.shift[co] : 0.f;
  stamp(1);
  // pipeline: global -> registers (3 steps ahead) -> LDS ring of 3 (2 steps ahead) -> MFMA; filters one sub-step ahead
  // the first two stages are requested together (a second register set, dead after the prologue) so that their
  // latencies overlap instead of adding up
  f32x4 g1[WNLD];
  gload(0);
  bload(0, grp);
  if (NS > 1) {
#pragma unroll
    for (int q = 0; q < WNLD; ++q) g1[q] = dc_bload4(xr, gofs[q], (unsigned)(WKC * 4));
  }
  stamp(2);
  sstore(0);
  if (NS > 2) gload(2);
  if (NS > 1) {
#pragma unroll
    for (int q = 0; q < WNLD; ++q) reinterpret_cast<f32x4*>(&stage[1][0])[sofs[q]] = g1[q];
  }
  stamp(3);
  if constexpr (NG == 1) {
    // one staged step: U = K % 3 is a compile-time constant so that the ring buffer offsets fold into the instructions
    auto step = [&](int K, auto u_tag) {
      constexpr int U = decltype(u_tag)::value;
      __syncthreads();  // buffers <= K+1 are complete; buffer (K+2)%3 is free
      lread(U, 0);
      bload(1, 2 * K + 1);
      compute(0);
      if (K + 2 < NS) sstore((U + 2) % 3);
      lread(U, 1);
      bload(0, 2 * K + 2 < 2 * NS ? 2 * K + 2 : 0);  // the tail load is a harmless re-read of step 0
      compute(1);
      if (K + 3 < NS) gload(K + 3);
    };
    for (int K0 = 0; K0 < NS; K0 += 3) {
      step(K0, std::integral_constant<int, 0>{});
      if (K0 + 1 < NS) step(K0 + 1, std::integral_constant<int, 1>{});
      if (K0 + 2 < NS) step(K0 + 2, std::integral_constant<int, 2>{});
    }
  } else {
    // group g computes sub-step g of every staged step; U = K % 3 (ring slot) and S = K % 2 (filter-fragment slot) are
    // compile-time constants: six steps per round of the loop
    auto step = [&](int K, auto u_tag, auto s_tag) {
      constexpr int U = decltype(u_tag)::value, S = decltype(s_tag)::value;
      __syncthreads();
      lread(U, grp);
      bload(S ^ 1, K + 1 < NS ? 2 * (K + 1) + grp : 0);  // the tail load is a harmless re-read of step 0
      compute(S);
      if (K + 2 < NS) sstore((U + 2) % 3);
      if (K + 3 < NS) gload(K + 3);
    };
    for (int K0 = 0; K0 < NS; K0 += 6) {
      step(K0, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
      if (K0 + 1 < NS) step(K0 + 1, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
      if (K0 + 2 < NS) step(K0 + 2, std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{});
      if (K0 + 3 < NS) step(K0 + 3, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
      if (K0 + 4 < NS) step(K0 + 4, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
      if (K0 + 5 < NS) step(K0 + 5, std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{});
    }
  }
  stamp(4);
  // inverse transform: over j in registers (P[b] = sum_j M[i][j] A[j][b]), over i (and the two wave groups) through LDS
  __syncthreads();  // every wave is done reading the staging ring, which the partials now overwrite
#pragma unroll
  for (int r4 = 0; r4 < 4; ++r4) {
    part[grp][i][0][tf][r4][lane] = acc[0][r4] + acc[1][r4] + acc[2][r4];
    part[grp][i][1][tf][r4][lane] = acc[1][r4] - acc[2][r4] - acc[3][r4];
  }
  __syncthreads();
  stamp(5);
  const int a = (wave >> 1) & 1, bq = wave & 1;  // this wave finalises output pixel (a, bq) of the tiles of fragment tf
  float* yb = reinterpret_cast<float*>(p.y);
  const float* rbp = reinterpret_cast<const float*>(p.resid);
  stamp(6);
  // NG = 2: the sixteen waves split the four accumulator rows (group g finalises rows 2g, 2g+1) and add the two groups' partial
  // sums, group 0's first: the order of every sum is fixed, the result does not depend on which wave arrives when
#pragma unroll
  for (int rr = 0; rr < 4 / NG; ++rr) {
    const int r4 = NG == 1 ? rr : 2 * grp + rr;
    float p0 = part[0][0][bq][tf][r4][lane], p1 = part[0][1][bq][tf][r4][lane], p2 = part[0][2][bq][tf][r4][lane], p3 = part[0][3][bq][tf][r4][lane];
    if constexpr (NG == 2) {
      p0 += part[NG - 1][0][bq][tf][r4][lane], p1 += part[NG - 1][1][bq][tf][r4][lane];
      p2 += part[NG - 1][2][bq][tf][r4][lane], p3 += part[NG - 1][3][bq][tf][r4][lane];
    }
    float v = a == 0 ? p0 + p1 + p2 : p1 - p2 - p3;
    const int q = 4 * (lane >> 4) + r4;  // D layout: row (tile in fragment) = 4*(lane/16) + r, col (channel) = lane%16
    const int ty = by * WBTY + 2 * tf + (q >> 3), tx = bx * WBTX + (q & 7);
    const int oy = phy + d * (2 * ty + a), ox = phx + d * (2 * tx + bq);
    if (oy < p.OH && ox < p.OW) {
      const long off = (long)n * p.y_img_stride + (long)oy * p.y_row_stride + (long)ox * p.y_pix_stride + co;
      v = v * sc + sh;
      if (rbp) v += rbp[off];
      if (p.relu) v = fmaxf(v, 0.f);
      yb[off] = v;
    }
  }
  stamp(7);
}

bool wino_eligible(const ConvGemmParams& p) {
  if (p.esize == 2) return wino_half_eligible(p);  // the float16 kernel (wino_f16.hip)
  const int d = p.ddy;  // dilation (1 or more), the same along x and y, with pad = dilation ("same" convolution)
  if (p.esize != 4 || p.nty != 3 || p.ntx != 3 || p.sy != 1 || d < 1 || d > 4 || p.dy0 != -d) return false;
  const int C = p.klen;
  if (C <= 0 || C % WKC != 0 || p.Cout % WBN != 0 || p.sigmoid_ch != 0) return false;
  if (p.sx != C || p.ddx != d * C || p.x0 != -d * C) return false;           // stride 1, dilation d, pad d along x
  if (p.x_rowlen % C != 0 || p.x_row_stride != p.x_rowlen) return false;     // dense NHWC rows of C channels
  if (p.OH != p.x_rows || p.OW != p.x_rowlen / C) return false;              // "same" convolution
  // 32-bit byte offsets (buffer addressing): one image of the input and the packed filter image stay below 2 GiB
  if ((long long)p.x_rows * p.x_row_stride * 4 >= 0x7fffffffLL || (long long)wino_packed_floats(p.Cout, C) * 4 >= 0x7fffffffLL) return false;
  return true;
}

long wino_grid(const ConvGemmParams& p) {
  if (p.esize == 2 && p.nty == 1 && p.ntx == 1) return stream1x1_grid(p);  // (a 1x1 layer on a form of its own: the streaming one)
  if (p.esize == 4 && p.nty == 1 && p.ntx == 1) return stream1x1f_grid(p);
  if (p.esize == 2 && p.nty == 7 && p.ntx == 1) return stem7x7_grid(p);    // (the stem)
  if (p.esize == 4 && p.nty == 7 && p.ntx == 1) return stem_ws_grid(p);
  if (p.esize == 2) return wino_half_grid(p);
  const int d = p.ddy;
  const int TY = ((p.OH + d - 1) / d + 1) / 2, TX = ((p.OW + d - 1) / d + 1) / 2;
  return (long)p.NB * d * d * ((TY + WBTY - 1) / WBTY) * ((TX + WBTX - 1) / WBTX) * (p.Cout / WBN);
}

size_t wino_packed_floats(int Cout, int Cin) { return (size_t)16 * Cout * Cin; }

void wino_pack_filters(const float* g, int Cout, int Cin, float* out) {
  static const double G[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
  for (int co = 0; co < Cout; ++co)
    for (int ci = 0; ci < Cin; ++ci) {
      const float* w = g + ((size_t)co * Cin + ci) * 9;
      double tmp[4][3], U[4][4];
      for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 3; ++b) tmp[a][b] = G[a][0] * w[b] + G[a][1] * w[3 + b] + G[a][2] * w[6 + b];
      for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b) U[a][b] = tmp[a][0] * G[b][0] + tmp[a][1] * G[b][1] + tmp[a][2] * G[b][2];
      const int nt = co / 16, col = co % 16, k16 = ci / 16, kg = (ci % 16) / 4, s = ci % 4;
      const int lane = kg * 16 + col;
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
          out[(((((size_t)nt * 4 + i) * (Cin / 16) + k16) * 4 + j) * 64 + lane) * 4 + s] = (float)U[i][j];
    }
}

const char* wino_variant_name(int variant) {
  return variant == kStemFloat ? "ws7x7f" : variant == kStreamFloat ? "ws1x1f" : variant == kStemHalf ? "stem7x7" : variant == kStreamHalf ? "ws1x1" : variant == kWinoHalf ? "wino_h23" : variant == kWinoVariant16 ? "wino_f23_w16" : "wino_f23";
}
const char* wino_kernel_label(int variant) {
  return variant == kStemFloat ? "ws7x7f<16x64>" : variant == kStreamFloat ? "ws1x1f<16xN>" : variant == kStemHalf ? "stem7x7<8x64>" : variant == kStreamHalf ? "ws1x1<32xN>" : variant == kWinoHalf ? "wino_h23<2x4x8x64>" : variant == kWinoVariant16 ? "wino_f23<4x8x16_w16>" : "wino_f23<4x8x16>";
}
int wino_variant_by_name(const char* name) {
  for (int v : {kWinoVariant, kWinoVariant16, kWinoHalf, kStreamHalf, kStemHalf, kStreamFloat, kStemFloat})
    if (name && std::strcmp(name, wino_variant_name(v)) == 0) return v;
  return -1;
}

int launch_wino_conv(const ConvGemmParams& p, void* stream, int variant) {
  if (variant == kStreamHalf) return launch_stream1x1(p, stream);
  if (variant == kStemHalf) return launch_stem7x7(p, stream);
  if (variant == kStreamFloat) return launch_stream1x1f(p, stream);
  if (variant == kStemFloat) return launch_stem_ws(p, stream);
  if (variant == kWinoHalf) return launch_wino_half(p, stream);
  if (p.esize != 4 || !wino_eligible(p) || !is_wino_variant(variant)) return (int)hipErrorInvalidValue;
  const long grid = wino_grid(p);
  if (grid <= 0) return 0;
  if (grid > 0x7fffffffL) return (int)hipErrorInvalidValue;
  ConvGemmParams q = p;
  static const int xcd_map = getenv("DC_XCD_MAP") ? atoi(getenv("DC_XCD_MAP")) : 1;
  q.xcd_on = xcd_map && grid >= 16;
  {
    const int d = p.ddy;
    q.w_TY = ((p.OH + d - 1) / d + 1) / 2, q.w_TX = ((p.OW + d - 1) / d + 1) / 2;
    q.w_NBY = (q.w_TY + WBTY - 1) / WBTY, q.w_NBX = (q.w_TX + WBTX - 1) / WBTX;
    q.w_nblk = p.NB * d * d * q.w_NBY * q.w_NBX;
    dc_magic((unsigned)q.w_nblk, q.w_div_nblk);
    dc_magic((unsigned)(q.w_NBY * q.w_NBX), q.w_div_nbyx);
    dc_magic((unsigned)(d * d), q.w_div_dd);
    dc_magic((unsigned)d, q.w_div_d);
    dc_magic((unsigned)q.w_NBX, q.w_div_nbx);
  }
  if (variant == kWinoVariant16) hipLaunchKernelGGL(wino_f23_kernel<2>, dim3((unsigned)grid), dim3(2 * WNTH), 0, (hipStream_t)stream, q);
  else hipLaunchKernelGGL(wino_f23_kernel<1>, dim3((unsigned)grid), dim3(WNTH), 0, (hipStream_t)stream, q);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// MAX pooling (NHWC, 16 bytes of channels per thread), windows clipped to the image
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void maxpool_kernel(const T* __restrict__ x, T* __restrict__ y, int NB, int H, int W,
                                                      int C, int OH, int OW, int k, int s, int pad) {
  constexpr int V = 16 / sizeof(T);
  typedef T vec_t __attribute__((ext_vector_type(V)));
  const int cvn = C / V;
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)NB * OH * OW * cvn;
  if (idx >= total) return;
  int cv = (int)(idx % cvn);
  long pix = idx / cvn;
  int ox = (int)(pix % OW);
  long t2 = pix / OW;
  int oy = (int)(t2 % OH);
  int n = (int)(t2 / OH);
  int hs = oy * s - pad, ws = ox * s - pad;
  int he = min(hs + k, H), we = min(ws + k, W);  // pooling_layer.cpp:150-155
  hs = max(hs, 0);
  ws = max(ws, 0);
  vec_t m;
#pragma unroll
  for (int q = 0; q < V; ++q) m[q] = (T)(sizeof(T) == 4 ? -3.402823466e+38f : -65504.f);
  for (int iy = hs; iy < he; ++iy)
    for (int ix = ws; ix < we; ++ix) {
      vec_t v = *reinterpret_cast<const vec_t*>(x + (((long)n * H + iy) * W + ix) * C + cv * V);
#pragma unroll
      for (int q = 0; q < V; ++q) m[q] = v[q] > m[q] ? v[q] : m[q];
    }
  *reinterpret_cast<vec_t*>(y + (((long)n * OH + oy) * OW + ox) * C + cv * V) = m;
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_scalar_kernel(const T* __restrict__ x, T* __restrict__ y, int NB, int H,
                                                             int W, int C, int OH, int OW, int k, int s, int pad) {
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
      float v = (float)x[(((long)n * H + iy) * W + ix) * C + c];
      m = v > m ? v : m;
    }
  y[idx] = (T)m;
}

template <typename T>
static int launch_maxpool_t(const void* x, void* y, int NB, int H, int W, int C, int OH, int OW, int k, int s, int pad,
                            void* stream) {
  constexpr int V = 16 / sizeof(T);
  if (C % V == 0) {
    long total = (long)NB * OH * OW * (C / V);
    if (total <= 0) return 0;
    hipLaunchKernelGGL(maxpool_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const T*)x, (T*)y, NB, H, W, C, OH, OW, k, s, pad);
  } else {
    long total = (long)NB * OH * OW * C;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(maxpool_scalar_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, (const T*)x, (T*)y, NB, H, W, C, OH, OW, k, s, pad);
  }
  return (int)hipGetLastError();
}

int launch_maxpool(const void* x, void* y, int esize, int NB, int H, int W, int C, int OH, int OW, int k, int s,
                   int pad, void* stream) {
  return esize == 2 ? launch_maxpool_t<_Float16>(x, y, NB, H, W, C, OH, OW, k, s, pad, stream)
                    : launch_maxpool_t<float>(x, y, NB, H, W, C, OH, OW, k, s, pad, stream);
}

// ------------------------------------------------------------------------------------------------
// stand-alone elementwise: y = act(x*a[c] + b[c] + z)   (arithmetic in float)
// ------------------------------------------------------------------------------------------------
__device__ inline float dc_act(float v, int relu, int sigmoid) {
  if (relu) v = fmaxf(v, 0.f);
  if (sigmoid) v = 1.f / (1.f + expf(-v));
  return v;
}

template <typename T>
__global__ __launch_bounds__(256) void eltwise_vec_kernel(const T* __restrict__ x, const T* __restrict__ z,
                                                          const float* __restrict__ a, const float* __restrict__ b,
                                                          T* __restrict__ y, long totalv, int C, int relu, int sigmoid) {
  constexpr int V = 16 / sizeof(T);
  typedef T vec_t __attribute__((ext_vector_type(V)));
  long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < totalv; i += stride) {
    vec_t v = reinterpret_cast<const vec_t*>(x)[i];
    vec_t zz;
    if (z) zz = reinterpret_cast<const vec_t*>(z)[i];
    int c = (int)((i * V) % C);
    vec_t o;
#pragma unroll
    for (int q = 0; q < V; ++q) {
      float f = (float)v[q];
      if (a) f *= a[c + q];
      if (b) f += b[c + q];
      if (z) f += (float)zz[q];
      o[q] = (T)dc_act(f, relu, sigmoid);
    }
    reinterpret_cast<vec_t*>(y)[i] = o;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void eltwise_scalar_kernel(const T* __restrict__ x, const T* __restrict__ z,
                                                             const float* __restrict__ a, const float* __restrict__ b,
                                                             T* __restrict__ y, long total, int C, int relu,
                                                             int sigmoid) {
  long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    float v = (float)x[i];
    int c = (int)(i % C);
    if (a) v *= a[c];
    if (b) v += b[c];
    if (z) v += (float)z[i];
    y[i] = (T)dc_act(v, relu, sigmoid);
  }
}

template <typename T>
static int launch_eltwise_t(const void* x, const void* z, const float* a, const float* b, void* y, long total, int C,
                            int relu, int sigmoid, void* stream) {
  constexpr int V = 16 / sizeof(T);
  if (total <= 0) return 0;
  if (C % V == 0 && total % V == 0) {
    long tv = total / V;
    long blocks = (tv + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(eltwise_vec_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const T*)x,
                       (const T*)z, a, b, (T*)y, tv, C, relu, sigmoid);
  } else {
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(eltwise_scalar_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       (const T*)x, (const T*)z, a, b, (T*)y, total, C, relu, sigmoid);
  }
  return (int)hipGetLastError();
}

int launch_eltwise(const void* x, const void* z, const float* a, const float* b, void* y, int esize, long total, int C,
                   int relu, int sigmoid, void* stream) {
  return esize == 2 ? launch_eltwise_t<_Float16>(x, z, a, b, y, total, C, relu, sigmoid, stream)
                    : launch_eltwise_t<float>(x, z, a, b, y, total, C, relu, sigmoid, stream);
}

// ------------------------------------------------------------------------------------------------
// crop (NHWC)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void crop_kernel(const T* __restrict__ x, T* __restrict__ y, int NB, int H, int W,
                                                   int C, int oh, int ow, int OH, int OW) {
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

int launch_crop(const void* x, void* y, int esize, int NB, int H, int W, int C, int oh, int ow, int OH, int OW,
                void* stream) {
  long total = (long)NB * OH * OW * C;
  if (total <= 0) return 0;
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (esize == 2)
    hipLaunchKernelGGL(crop_kernel<_Float16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       (const _Float16*)x, (_Float16*)y, NB, H, W, C, oh, ow, OH, OW);
  else
    hipLaunchKernelGGL(crop_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float*)x,
                       (float*)y, NB, H, W, C, oh, ow, OH, OW);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// NCHW float (Blob side) <-> NHWC float / half (device image) through a 32x32 LDS tile, both sides coalesced
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int C,
                                                           int HW, int CP) {
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
    if (pix < HW && c < CP) dst[((long)n * HW + pix) * CP + c] = (T)tile[tx][ty + 8 * k];
  }
}

template <typename T, typename D>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const T* __restrict__ src, D* __restrict__ dst, int C,
                                                           int HW, int CP, int cbase) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z;
  const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int pix = p0 + ty + 8 * k, c = c0 + tx;
    tile[ty + 8 * k][tx] = (pix < HW && c < C) ? (float)src[((long)n * HW + pix) * CP + cbase + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int c = c0 + ty + 8 * k, pix = p0 + tx;
    if (c < C && pix < HW) dst[((long)n * C + c) * HW + pix] = (D)tile[tx][ty + 8 * k];
  }
}

int launch_nchw_to_nhwc(const float* src, void* dst, int esize, int NB, int C, int H, int W, int CP, void* stream) {
  int HW = H * W;
  if (NB <= 0 || HW <= 0) return 0;
  dim3 grid((HW + 31) / 32, (CP + 31) / 32, NB);
  if (esize == 2)
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<_Float16>, grid, dim3(256), 0, (hipStream_t)stream, src, (_Float16*)dst, C, HW, CP);
  else
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, src, (float*)dst, C, HW, CP);
  return (int)hipGetLastError();
}

int launch_nhwc_to_nchw(const void* src, void* dst, int esize, int NB, int C, int H, int W, int CP, int c0,
                        void* stream, int dst_esize) {
  int HW = H * W;
  if (NB <= 0 || HW <= 0) return 0;
  if ((dst_esize != 2 && dst_esize != 4) || (dst_esize == 2 && esize != 2)) return (int)hipErrorInvalidValue;  // f32 image -> f16 copy: not offered
  dim3 grid((HW + 31) / 32, (C + 31) / 32, NB);
  if (esize == 2 && dst_esize == 2)
    hipLaunchKernelGGL((nhwc_to_nchw_kernel<_Float16, _Float16>), grid, dim3(256), 0, (hipStream_t)stream, (const _Float16*)src, (_Float16*)dst, C, HW, CP, c0);
  else if (esize == 2)
    hipLaunchKernelGGL((nhwc_to_nchw_kernel<_Float16, float>), grid, dim3(256), 0, (hipStream_t)stream, (const _Float16*)src, (float*)dst, C, HW, CP, c0);
  else
    hipLaunchKernelGGL((nhwc_to_nchw_kernel<float, float>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)src, (float*)dst, C, HW, CP, c0);
  return (int)hipGetLastError();
}

// float -> half conversion of a packed filter image (upload path of fp16 nets)
__global__ __launch_bounds__(256) void f32_to_f16_kernel(const float* __restrict__ src, _Float16* __restrict__ dst, long n) {
  long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = (_Float16)src[i];
}
int launch_f32_to_f16(const float* src, void* dst, long n, void* stream) {
  if (n <= 0) return 0;
  long blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(f32_to_f16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, (_Float16*)dst, n);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// pose decode on the device (estimate_pose.py:131-143 `_pose_from_mats`): per joint the FIRST maximum of
// the score map in row-major order, refined by the location-regression vector at that cell.  One block
// per (image, joint); only 5 x J doubles per image leave the GPU instead of the maps.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void pose_decode_kernel(const T* __restrict__ prob, int pcp, int pc0,
                                                          const T* __restrict__ loc, int lcp, int lc0, int H, int W,
                                                          int J, double scale, double* __restrict__ out) {
  __shared__ float sv[256];
  __shared__ int si[256];
  const int j = blockIdx.x, n = blockIdx.y, HW = H * W;
  float best = -3.402823466e+38f;
  int bi = 0x7fffffff;
  for (int p = threadIdx.x; p < HW; p += 256) {
    const float v = (float)prob[((long)n * HW + p) * pcp + pc0 + j];
    if (v > best) best = v, bi = p;  // strided scan keeps the smallest index per thread
  }
  sv[threadIdx.x] = best;
  si[threadIdx.x] = bi;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      const float v2 = sv[threadIdx.x + s];
      const int i2 = si[threadIdx.x + s];
      if (v2 > sv[threadIdx.x] || (v2 == sv[threadIdx.x] && i2 < si[threadIdx.x])) {
        sv[threadIdx.x] = v2;
        si[threadIdx.x] = i2;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int p = si[0] == 0x7fffffff ? 0 : si[0];
    const int row = p / W, col = p - row * W;
    const double kLoc = 7.280109889280518;  // sqrt(53)  (_LOCREF_SCALE_MUL, estimate_pose.py:27)
    const double ox = (double)(float)loc[((long)n * HW + p) * lcp + lc0 + 2 * j];
    const double oy = (double)(float)loc[((long)n * HW + p) * lcp + lc0 + 2 * j + 1];
    double* o = out + (long)n * 5 * J;
    o[0 * J + j] = ((double)col * 8.0 + 4.0 + ox * kLoc) / scale;
    o[1 * J + j] = ((double)row * 8.0 + 4.0 + oy * kLoc) / scale;
    o[2 * J + j] = (double)(float)prob[((long)n * HW + p) * pcp + pc0 + j];
    o[3 * J + j] = oy * kLoc / scale;
    o[4 * J + j] = ox * kLoc / scale;
  }
}

int launch_pose_decode(const void* prob, int pcp, int pc0, const void* loc, int lcp, int lc0, int esize, int NB, int H,
                       int W, int J, double scale, double* out, void* stream) {
  if (NB <= 0 || J <= 0) return 0;
  if (esize == 2)
    hipLaunchKernelGGL(pose_decode_kernel<_Float16>, dim3(J, NB), dim3(256), 0, (hipStream_t)stream, (const _Float16*)prob,
                       pcp, pc0, (const _Float16*)loc, lcp, lc0, H, W, J, scale, out);
  else
    hipLaunchKernelGGL(pose_decode_kernel<float>, dim3(J, NB), dim3(256), 0, (hipStream_t)stream, (const float*)prob, pcp,
                       pc0, (const float*)loc, lcp, lc0, H, W, J, scale, out);
  return (int)hipGetLastError();
}

// ---- multi-person consumers: part candidates (NMS) and pairwise regression decode ------------------------------------
// One workgroup per (image, joint) score map.  Every cell is tested for being the maximum of its (2r+1)^2 window (ties: the
// lower cell index wins) and, if so, becomes the 64-bit key (score bits << 32 | ~cell): keys are unique, and descending key
// order IS the output order (score descending, cell ascending).  The candidate SET does not depend on thread timing, and
// the list is then ordered by the whole workgroup: up to kPartLds keys by a bitonic sort in LDS; a map with more local
// maxima than that (threshold 0, radius 0) spills its keys to global memory and takes the first max_det by repeated
// workgroup-wide maximum.  Nothing is dropped in arrival order, so the result is deterministic for every input.
constexpr int kPartLds = 4096;  // keys sorted in LDS (32 KB)

template <typename T>
__global__ __launch_bounds__(256) void part_select_kernel(const T* __restrict__ prob, int pcp, int pc0, const T* __restrict__ loc, int lcp,
                                                          int lc0, int H, int W, int J, float thr, int radius, double scale, int max_det,
                                                          unsigned long long* __restrict__ spill, int* __restrict__ counts,
                                                          double* __restrict__ out) {
  __shared__ unsigned long long keys[kPartLds];
  __shared__ unsigned long long red[256];
  __shared__ int cnt;
  const int nj = blockIdx.x, n = nj / J, j = nj - n * J, t = threadIdx.x, HW = H * W;
  const T* base = prob + ((long)n * HW) * pcp + pc0 + j;
  unsigned long long* mine = spill + (long)nj * HW;
  if (t == 0) cnt = 0;
  __syncthreads();
  for (int cell = t; cell < HW; cell += 256) {
    const int row = cell / W, col = cell - row * W;
    const float v = (float)base[(long)cell * pcp];
    bool ok = v >= thr;
    for (int dy = -radius; ok && dy <= radius; ++dy) {
      const int y = row + dy;
      if (y < 0 || y >= H) continue;
      for (int dx = -radius; dx <= radius; ++dx) {
        const int x = col + dx;
        if (x < 0 || x >= W || (dy == 0 && dx == 0)) continue;
        const float u = (float)base[((long)y * W + x) * pcp];
        if (u > v || (u == v && y * W + x < cell)) {
          ok = false;
          break;
        }
      }
    }
    if (ok) {
      // the raw bit pattern orders NON-NEGATIVE floats only (a set sign bit would sort above every positive score, in reverse):
      // v >= thr >= 0 here — Net::detect_parts refuses a negative threshold — and key 0 (the padding) is below every candidate
      const unsigned long long key = ((unsigned long long)__float_as_uint(v) << 32) | (unsigned)(0xffffffffu - (unsigned)cell);
      const int slot = atomicAdd(&cnt, 1);  // LDS counter: the slot order varies, the set and (after sorting) the result do not
      if (slot < kPartLds) keys[slot] = key;
      mine[slot] = key;
    }
  }
  __syncthreads();
  const int m = cnt;
  const int take = min(m, max_det);
  if (t == 0) counts[nj] = take;
  double* o = out + (long)nj * max_det * 5;
  auto emit = [&](int k, unsigned long long key) {
    const double kLoc = 7.280109889280518;  // sqrt(53)
    const int cell = (int)(0xffffffffu - (unsigned)key);
    const int row = cell / W, col = cell - row * W;
    const T* l = loc + (((long)n * H + row) * W + col) * lcp + lc0 + 2 * j;
    double* q = o + (long)k * 5;
    q[0] = ((double)col * 8.0 + 4.0 + (double)(float)l[0] * kLoc) / scale;
    q[1] = ((double)row * 8.0 + 4.0 + (double)(float)l[1] * kLoc) / scale;
    q[2] = (double)__uint_as_float((unsigned)(key >> 32));
    q[3] = (double)row;
    q[4] = (double)col;
  };
  for (int k = take + t; k < max_det; k += 256) {
    double* q = o + (long)k * 5;
    q[0] = q[1] = q[2] = 0.0;
    q[3] = q[4] = -1.0;
  }
  if (m <= kPartLds) {
    int P = 1;
    while (P < m) P <<= 1;
    for (int i = m + t; i < P; i += 256) keys[i] = 0ull;  // below every real key (scores >= 0, cell term > 0)
    __syncthreads();
    for (int k2 = 2; k2 <= P; k2 <<= 1)
      for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
        for (int i = t; i < P; i += 256) {
          const int ixj = i ^ j2;
          if (ixj > i) {
            const unsigned long long a = keys[i], b = keys[ixj];
            const bool desc = (i & k2) == 0;  // descending overall
            if (desc ? a < b : a > b) keys[i] = b, keys[ixj] = a;
          }
        }
        __syncthreads();
      }
    for (int k = t; k < take; k += 256) emit(k, keys[k]);
  } else {
    __threadfence_block();
    unsigned long long prev = ~0ull;
    for (int k = 0; k < take; ++k) {  // k-th largest key = the largest key below the previous one
      unsigned long long best = 0ull;
      for (int i = t; i < m; i += 256) {
        const unsigned long long v = mine[i];
        if (v < prev && v > best) best = v;
      }
      red[t] = best;
      __syncthreads();
      for (int s2 = 128; s2 > 0; s2 >>= 1) {
        if (t < s2 && red[t + s2] > red[t]) red[t] = red[t + s2];
        __syncthreads();
      }
      prev = red[0];
      if (t == 0) emit(k, prev);
      __syncthreads();
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void pairwise_decode_kernel(const T* __restrict__ next, int ncp, int nc0, int NB, int H, int W, int E,
                                                              double scale, int ndet, const int* __restrict__ det,
                                                              const double* __restrict__ mean, const double* __restrict__ stdev,
                                                              double* __restrict__ out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)ndet * E) return;
  const int d = (int)(i / E), l = (int)(i - (long)d * E);
  const int n = det[3 * d], row = det[3 * d + 1], col = det[3 * d + 2];
  double* o = out + i * 2;
  if (n < 0 || n >= NB || row < 0 || row >= H || col < 0 || col >= W) {
    o[0] = o[1] = 0.0;
    return;
  }
  const T* p = next + (((long)n * H + row) * W + col) * ncp + nc0 + 2 * l;
  const double m0 = mean ? mean[2 * l] : 0.0, m1 = mean ? mean[2 * l + 1] : 0.0;
  const double s0 = stdev ? stdev[2 * l] : 1.0, s1 = stdev ? stdev[2 * l + 1] : 1.0;
  o[0] = ((double)col * 8.0 + 4.0 + (double)(float)p[0] * s0 + m0) / scale;
  o[1] = ((double)row * 8.0 + 4.0 + (double)(float)p[1] * s1 + m1) / scale;
}

int launch_part_select(const void* prob, int pcp, int pc0, const void* loc, int lcp, int lc0, int esize, int NB, int H, int W, int J, float thr,
                       int radius, double scale, int max_det, unsigned long long* spill, int* counts, double* out, void* stream) {
  if (NB * J <= 0) return 0;
  if (esize == 2)
    hipLaunchKernelGGL(part_select_kernel<_Float16>, dim3(NB * J), dim3(256), 0, (hipStream_t)stream, (const _Float16*)prob, pcp, pc0,
                       (const _Float16*)loc, lcp, lc0, H, W, J, thr, radius, scale, max_det, spill, counts, out);
  else
    hipLaunchKernelGGL(part_select_kernel<float>, dim3(NB * J), dim3(256), 0, (hipStream_t)stream, (const float*)prob, pcp, pc0,
                       (const float*)loc, lcp, lc0, H, W, J, thr, radius, scale, max_det, spill, counts, out);
  return (int)hipGetLastError();
}

int launch_pairwise_decode(const void* next, int ncp, int nc0, int esize, int NB, int H, int W, int E, double scale, int ndet,
                           const int* det, const double* mean, const double* stdev, double* out, void* stream) {
  const long total = (long)ndet * E;
  if (total <= 0) return 0;
  const dim3 grid((unsigned)((total + 255) / 256));
  if (esize == 2)
    hipLaunchKernelGGL(pairwise_decode_kernel<_Float16>, grid, dim3(256), 0, (hipStream_t)stream, (const _Float16*)next, ncp, nc0, NB, H, W,
                       E, scale, ndet, det, mean, stdev, out);
  else
    hipLaunchKernelGGL(pairwise_decode_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)next, ncp, nc0, NB, H, W, E,
                       scale, ndet, det, mean, stdev, out);
  return (int)hipGetLastError();
}

// ---- image pre-processing -----------------------------------------------------------------------------------------
// Integer work at a few bytes per pixel: HBM/latency-bound, one thread per output pixel, no LDS.
__device__ __forceinline__ int clip8_fixed(int acc) {
  const int v = acc >> 22;  // PRECISION_BITS = 32 - 8 - 2 (Pillow Resample.c)
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

__global__ __launch_bounds__(256) void image_resample_x_kernel(ImagePrepParams p) {
  const long total = (long)p.n * p.rows * p.use_w;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int x = (int)(i % p.use_w);
  const long t = i / p.use_w;
  const int r = (int)(t % p.rows), n = (int)(t / p.rows);
  const int sy = min(p.row0 + r, p.h - 1);  // rows >= h replicate the last row (estimate_pose.py:89-92)
  const unsigned char* row = p.src + ((long)n * p.h + sy) * p.w * 3;
  const int xmin = p.x_bounds[2 * x], cnt = p.x_bounds[2 * x + 1];
  const int* kk = p.x_coeffs + (long)x * p.x_ksize;
  int a0 = 1 << 21, a1 = 1 << 21, a2 = 1 << 21;
  for (int k = 0; k < cnt; ++k) {
    const int sx = min(xmin + k, p.w - 1);  // columns >= w replicate the last column (:93-95)
    const int c = kk[k];
    a0 += row[sx * 3 + 0] * c;
    a1 += row[sx * 3 + 1] * c;
    a2 += row[sx * 3 + 2] * c;
  }
  reinterpret_cast<uchar4*>(p.tmp)[i] = make_uchar4((unsigned char)clip8_fixed(a0), (unsigned char)clip8_fixed(a1),
                                                    (unsigned char)clip8_fixed(a2), 0);
}

template <typename T>
__global__ __launch_bounds__(256) void image_finish_kernel(ImagePrepParams p) {
  const long total = (long)p.n * p.out_h * p.out_w;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int x = (int)(i % p.out_w);
  const long t = i / p.out_w;
  const int y = (int)(t % p.out_h), n = (int)(t / p.out_h);
  float v[3] = {0.f, 0.f, 0.f};
  if (y < p.use_h && x < p.use_w) {
    auto fetch = [&](int row, int& b0, int& b1, int& b2) {
      if (p.x_bounds) {
        const uchar4 q = reinterpret_cast<const uchar4*>(p.tmp)[((long)n * p.rows + (row - p.row0)) * p.use_w + x];
        b0 = q.x, b1 = q.y, b2 = q.z;
      } else {
        const unsigned char* s = p.src + (((long)n * p.h + min(row, p.h - 1)) * p.w + min(x, p.w - 1)) * 3;
        b0 = s[0], b1 = s[1], b2 = s[2];
      }
    };
    int o0, o1, o2;
    if (p.y_bounds) {
      const int ymin = p.y_bounds[2 * y], cnt = p.y_bounds[2 * y + 1];
      const int* kk = p.y_coeffs + (long)y * p.y_ksize;
      int a0 = 1 << 21, a1 = 1 << 21, a2 = 1 << 21;
      for (int k = 0; k < cnt; ++k) {
        int b0, b1, b2;
        fetch(ymin + k, b0, b1, b2);
        const int c = kk[k];
        a0 += b0 * c, a1 += b1 * c, a2 += b2 * c;
      }
      o0 = clip8_fixed(a0), o1 = clip8_fixed(a1), o2 = clip8_fixed(a2);
    } else {
      fetch(y, o0, o1, o2);
    }
    v[0] = (float)o0 - p.mean[0], v[1] = (float)o1 - p.mean[1], v[2] = (float)o2 - p.mean[2];
  }
  T* d = reinterpret_cast<T*>(p.dst) + i * p.dst_cp;
  for (int c = 0; c < p.dst_cp; ++c) d[c] = (T)(c < 3 ? v[c] : 0.f);
}

int launch_image_prep(const ImagePrepParams& p, void* stream) {
  if (p.dst_esize != 2 && p.dst_esize != 4) return (int)hipErrorInvalidValue;
  if (p.dst_cp < 3 || p.use_h > p.out_h || p.use_w > p.out_w) return (int)hipErrorInvalidValue;
  if (p.x_bounds) {
    const long total = (long)p.n * p.rows * p.use_w;
    if (total > 0)
      hipLaunchKernelGGL(image_resample_x_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
  }
  const long total = (long)p.n * p.out_h * p.out_w;
  if (total <= 0) return 0;
  if (p.dst_esize == 2)
    hipLaunchKernelGGL(image_finish_kernel<_Float16>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL(image_finish_kernel<float>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
  return (int)hipGetLastError();
}

}  // namespace dc
