// kernels.h — launch interface of the gfx950 kernels (kernels.hip).  Plain structs, no HIP types
// in the signatures except the opaque stream, so the graph runtime (net_*.cpp) stays host-only C++.
#pragma once
#include <cstddef>
#include <cstdint>

namespace dc {

constexpr int kMaxTaps = 32;  // tap-validity masks are one 32-bit word per staged row

// out[pixel][co] = act( (sum_k A[pixel][k] * W[co][k]) * scale[co] + shift[co] (+ resid[pixel][co]) )
//   pixel = (n, oy, ox) over an NB x OH x OW grid,
//   A[pixel][.] = concatenation over a (nty x ntx) grid of TAPS of `klen` consecutive floats of source row
//                 oy*sy + dy0 + ty*ddy, starting at element ox*sx + x0 + tx*ddx  (0 outside the row/image),
//   W packed [Cout][Ktot] with k contiguous, taps in (ty, tx) order (Ktot = nty*ntx*klen).
// Covers every Convolution of the path (1x1, 1x1 stride 2, 3x3, dilated 3x3, the 7x7 stem seen as
// 7 row-taps of 8 NHWC4 pixels) and, per output-parity class, the stride-2 Deconvolution heads.
// The tap grid is arithmetic so the kernel advances it with scalar adds (no table loads in the K loop).
constexpr int kMaxClasses = 4;
// One output-residue class of a strided Deconvolution inside a MULTI-CLASS launch: the classes of a stride-s transposed
// convolution are s*s ordinary gather-GEMMs over the same input that differ only in tap grid, K, filter image and the
// interleaved output pixels they own, so they run as ONE launch (class = a range of the grid, heaviest class first so
// that its workgroups start first).  Fields mean what the same-named ConvGemmParams fields mean.
struct ConvClass {
  int nty, ntx, dy0, ddy, x0, ddx, Ktot, x_bias;
  int OH, OW, M;
  int tiles_m;          // m tiles of the class (filled by launch_conv_gemm)
  unsigned div_ohw[2];  // (filled by launch_conv_gemm)
  unsigned div_ow[2];
  long w_off;           // element offset of the class's filter image inside `w`
  long y_off;           // element offset of the class's first output inside `y` (and `resid`)
};

// MULTI-PROBLEM launches (round 4): the SAME layer over several tensors in one launch — the four scales of an image pyramid
// (python/pose/estimate_pose.py:81-128 runs them as four forwards; base_conv_layer.cpp:326-341 as one SGEMM per image), the
// crops of a crowd image, and, inside each of them, the residue classes of a strided deconvolution.  Filters, epilogue
// constants, klen, sy/sx, Cout are the layer's and stay in ConvGemmParams; everything that depends on a tensor's shape or
// address is per problem.  The table travels in the kernel arguments (ConvMultiArgs): filters are pulled through the L2s once
// per layer instead of once per scale, one dispatch ramp and one tail per layer instead of four.
constexpr int kMaxProblems = 16;
struct ConvProblem {
  const void* x;
  void* y;            // pre-offset to the first output element of the problem
  const void* resid;  // same addressing as y, or null (all problems of a launch alike)
  long x_img_stride, y_img_stride;
  long w_off;         // element offset of the problem's filter image inside `w` (deconvolution classes)
  int x_row_stride, x_rows, x_rowlen;
  int nty, ntx, dy0, ddy, x0, ddx, Ktot, x_bias;
  int OH, OW, M;
  int tiles_m;          // (filled by prepare_conv_multi)
  unsigned div_ohw[2];  // (filled)
  unsigned div_ow[2];   // (filled)
  int y_row_stride, y_pix_stride;
  int dense_x, dense_y;  // (filled) as ConvGemmParams::dense_x / dense_y
  int NB;
  int pad_;
};
struct ConvMultiTable {
  // XCD row qy of the (1 << mc_lgx) x (8 >> mc_lgx) arrangement walks, of EVERY problem k in turn, the m tiles
  // [tiles_m[k]*qy/gy, tiles_m[k]*(qy+1)/gy): end[qy][k] = m tiles of problems 0..k in that walk (INT_MAX past the last
  // problem), so a workgroup finds its problem with 16 scalar compares on one 64-byte line
  int end[8][kMaxProblems];
  ConvProblem prob[kMaxProblems];
};

struct ConvGemmParams {
  int esize;          // bytes per activation / filter element: 4 (float) or 2 (_Float16); strides are in elements
  const void* x;
  long x_img_stride;  // elements between images
  int x_row_stride;   // elements between rows
  int x_rows;         // H of the source
  int x_rowlen;       // valid elements in a row (W*C)
  int sy, sx;         // source step per output pixel: rows / elements
  int nty, ntx;       // tap grid
  int dy0, ddy;       // source-row offset of tap row ty: dy0 + ty*ddy
  int x0, ddx;        // element offset of tap column tx: x0 + tx*ddx
  int klen;           // K elements per tap (multiple of the variant's BK)
  int x_bias;         // min over taps of (dy*x_row_stride + xoff) (<= 0); filled by launch_conv_gemm
  const void* w;
  int Ktot;
  int NB, OH, OW;
  int M;  // NB*OH*OW
  int Cout;
  void* y;  // pre-offset to the first output element of this launch
  long y_img_stride;
  int y_row_stride;  // elements per oy step
  int y_pix_stride;  // elements per ox step
  const void* resid;  // same addressing as y (may alias y), or null
  const float* scale;  // [Cout] or null (=1)
  const float* shift;  // [Cout] or null (=0)
  int relu;
  int sigmoid_ch;  // channels [0, sigmoid_ch) get the logistic
  // --- filled by launch_conv_gemm (host-side precomputation keeps integer divisions out of the prologue) ---
  int xcd_on;              // XCD-aware tile map in use
  int xcd_lgx;             // the 8 XCDs form a (1 << xcd_lgx) x (8 >> xcd_lgx) arrangement over (n tiles) x (m tiles): XCD (qx, qy)
                           // owns n tiles [tn*qx/gx, tn*(qx+1)/gx) and m tiles [tm*qy/gy, tm*(qy+1)/gy) — computed with shifts in
                           // the kernel (a per-XCD table in the argument block cost every workgroup a second, dependent
                           // kernel-argument fetch: ~0.5 us before its first useful instruction)
  int tiles_m;
  unsigned div_rw[2][2];   // magic numbers for a division by the rectangle width: floor(tn/gx) and that + 1
  unsigned div_ohw[2];     // magic {multiplier, shift} for m / (OH*OW)
  unsigned div_ow[2];      // ... for m / OW
  unsigned div_tn[2];      // ... for block / tiles_n (linear map)
  int tiles_n;
  long long* dbg;  // optional [grid][4 waves][6] device timestamps (DC_DEBUG_TIMING), else null
  // --- multi-class launches (the stride-2 deconvolution heads): ncls > 1 and cls[0..ncls) replace the single-problem
  //     fields nty..x_bias / Ktot / OH / OW / M above; sy, sx, klen, strides, Cout, epilogue are common to all classes
  // --- Winograd launches (filled by launch_wino_conv): tile-grid geometry and the magic numbers of its divisions
  int w_TY, w_TX, w_NBY, w_NBX, w_nblk;
  unsigned w_div_nblk[2], w_div_nbyx[2], w_div_dd[2], w_div_d[2], w_div_nbx[2];
  int wide_epi;  // (filled by launch_conv_gemm) float16: 16-byte epilogue through LDS — Cout and the output strides are multiples of 8
  int vec_epi;   // (filled) 16-byte output vectors are legal: Cout, the output strides and the bases are multiples of 16 bytes, no sigmoid
  int dense_x;   // (filled) 1x1 / stride-1 layer over a dense NHWC tensor: output pixel m reads the klen elements at m * sx (no decode)
  int dense_y;   // (filled) the output pixels are dense: pixel m is written at m * y_pix_stride
  int ncls;
  int mc_lgx;  // multi-class tile map: the 8 XCDs form a (1 << mc_lgx) x (8 >> mc_lgx) grid over (n tiles) x (m tiles of every class)
  ConvClass cls[kMaxClasses];
  // --- multi-problem launches: nprob > 0 and the ConvMultiTable that follows the block in the kernel arguments
  //     (ConvMultiArgs) replace every per-tensor field above
  int nprob;
  int pad2_;
};
// kernel arguments of a multi-problem launch: the table travels IN the argument block (3.6 KB of the 4 KB a HIP kernel may
// take), so a workgroup finds its problem with scalar loads from the same segment as everything else — with the table behind
// a pointer in device memory every workgroup paid one more dependent round trip (argument block -> table row -> problem)
// before its first useful instruction, ~5 % of a batch-8 float16 forward
struct ConvMultiArgs {
  ConvGemmParams p;
  ConvMultiTable t;
};
static_assert(sizeof(ConvMultiArgs) <= 4096, "HIP kernel arguments are limited to 4 KB");

// Tile variants of conv_gemm.  BM x BN output tile per 256-thread workgroup, 4 waves arranged
// WR x WC x WK (WK = waves splitting the K range of the same output tile, reduced through LDS).
struct ConvVariant {
  const char* name;
  int BM, BN, WR, WC, WK;
};
int conv_num_variants();
const ConvVariant& conv_variant(int i);
// workgroups this variant launches for the problem
int conv_variant_bk(int i);
int conv_variant_esize(int i);
bool conv_variant_multiclass(int i);  // has a multi-class instantiation (ConvGemmParams::ncls > 1)
long conv_grid(const ConvGemmParams& p, int variant);
// returns hipError_t as int
int launch_conv_gemm(const ConvGemmParams& p, int variant, void* stream);
// Multi-problem launch, prepared once (host side): `p` carries the layer's common fields (esize, klen, sy, sx, w, Cout, scale,
// shift, relu, sigmoid_ch), `table.prob[0..nprob)` the per-tensor ones (pointers included).  Fills the derived fields of both
// (x_bias, magic numbers, tiles, dense / vector-epilogue flags, the XCD arrangement and table.end) and returns the grid, or
// -1 if this variant cannot take the launch.
long prepare_conv_multi(ConvGemmParams& p, ConvMultiTable& table, int nprob, int variant);
bool conv_variant_multiproblem(int i);
int launch_conv_multi(const ConvMultiArgs& a, int variant, long grid, void* stream);

// ---- Winograd F(2x2, 3x3) for the stride-1, dilation-1, pad-1 3x3 convolutions (float32) -------------------------
// Same ConvGemmParams as the gather-GEMM (x/y/resid/scale/shift/relu, NB, OH, OW, Cout, strides; klen = input channels,
// x_rows = H, x_rowlen = W*klen); `w` is the transformed-filter image made by wino_pack_filters().
constexpr int kWinoVariant = 1000;             // Launch::variant value that selects this kernel: 8 waves per workgroup ("wino_f23")
constexpr int kWinoVariant16 = 1001;           // ... its 16-wave form ("wino_f23_w16": launches of at most one workgroup per CU, kernels.hip)
constexpr int kWinoHalf = 1002;                // the float16 kernel (wino_f16.hip, "wino_h23"): fp16 operands, fp32 accumulate
constexpr int kStreamHalf = 1003;              // NOT Winograd: the float16 streaming form of the dense 1x1 layers (stream1x1.hip, "ws1x1") — listed
                                               // here because it is handled like one everywhere: a form outside the tile table with a filter
                                               // image of its own (Launch::wino_w), timed against the tiles per shape
constexpr int kStemHalf = 1004;                // NOT Winograd either: the float16 7x7 / stride-2 stem as a kernel of its own (stem_f16.hip, "stem7x7")
constexpr int kStreamFloat = 1005;             // ... and the float32 form of the streaming 1x1 kernel (stream1x1_f32.hip, "ws1x1f")
constexpr int kStemFloat = 1006;               // ... and the float32 7x7 / stride-2 stem on that kernel's skeleton (stream1x1_f32.hip, "ws7x7f")
inline bool is_wino_variant(int v) { return v == kWinoVariant || v == kWinoVariant16 || v == kWinoHalf || v == kStreamHalf || v == kStemHalf || v == kStreamFloat || v == kStemFloat; }
inline int wino_variant_esize(int v) { return v == kWinoHalf || v == kStreamHalf || v == kStemHalf ? 2 : 4; }  // element size of the nets the form serves
const char* wino_variant_name(int variant);    // the tile name of tune caches / reports / set_tile
const char* wino_kernel_label(int variant);    // the kernel column of plan texts
int wino_variant_by_name(const char* name);    // -1: not a Winograd tile name
bool wino_eligible(const ConvGemmParams& p);   // geometry / type the kernel takes
long wino_grid(const ConvGemmParams& p);
size_t wino_packed_floats(int Cout, int Cin);
// g: [Cout][Cin][3][3] (Caffe order) -> U = G g G^T per (co, ci), laid out so that one wave's B-operand load is 1 KB
// contiguous: [Cout/16][4 i][Cin/16][4 j][64 lanes][4]
void wino_pack_filters(const float* g, int Cout, int Cin, float* out);
int launch_wino_conv(const ConvGemmParams& p, void* stream, int variant = kWinoVariant);
// ---- the same for a float16 net (wino_f16.hip): `w` is the image made by wino_half_pack_filters() uploaded as _Float16, `scale`
// must carry the extra factors 4 (the staged pixels are pre-multiplied by 1/4) and row_scale[co]; no shortcut operand
bool wino_half_eligible(const ConvGemmParams& p);
long wino_half_grid(const ConvGemmParams& p);
size_t wino_half_packed_elems(int Cout, int Cin);
void wino_half_pack_filters(const float* g, int Cout, int Cin, bool rowscale, float* out, float* row_scale);
int launch_wino_half(const ConvGemmParams& p, void* stream);
// ---- the streaming form of the dense float16 1x1 / stride-1 layers (stream1x1.hip): filters resident in registers, the pixels walked in
// 32-pixel steps through an LDS-DMA ring.  `w` is the image made by stream1x1_pack_filters() uploaded as _Float16; x / y / resid / scale /
// shift / relu as the gather-GEMM (the epilogue is the same instruction sequence: bit-identical results)
bool stream1x1_eligible(const ConvGemmParams& p);
long stream1x1_grid(const ConvGemmParams& p);
size_t stream1x1_packed_elems(int Cout, int K);
void stream1x1_pack_filters(const float* g, int Cout, int K, float* out);
int launch_stream1x1(const ConvGemmParams& p, void* stream);
// ---- the float32 form (stream1x1_f32.hip): 16-pixel steps of v_mfma_f32_16x16x4_f32, one workgroup per CU (two at K <= 128); K = 64, 128, 256 or 512, Cout % 64 == 0
bool stream1x1f_eligible(const ConvGemmParams& p);
long stream1x1f_grid(const ConvGemmParams& p);
size_t stream1x1f_packed_elems(int Cout, int K);
void stream1x1f_pack_filters(const float* g, int Cout, int K, float* out);
int launch_stream1x1f(const ConvGemmParams& p, void* stream);
// ---- the float32 stem on the same skeleton ("ws7x7f"): conv1 7x7 / 2 over the NHWC4 image as the lowering's 7-row-tap launch describes it; a
// 1 KiB request gathers an output pixel's 7 x 8 input pixels (K = 224, the row-tap image's columns); `w` from stem_ws_pack_filters()
bool stem_ws_eligible(const ConvGemmParams& p);
long stem_ws_grid(const ConvGemmParams& p);
size_t stem_ws_packed_elems();
void stem_ws_pack_filters(const float* rowtap, float* out);  // rowtap: [64][224] = the row-tap image (k = ky 32 + kx 4 + ci)
int launch_stem_ws(const ConvGemmParams& p, void* stream);
// ---- the float16 stem (stem_f16.hip): conv1 7x7 / 2 over the NHWC4 image, as the lowering's 7-row-tap launch describes it (4 or 8 channels
// per pixel, at most 4 of them real); `w` is the image made by stem7x7_pack_filters(), uploaded as _Float16
bool stem7x7_eligible(const ConvGemmParams& p);
long stem7x7_grid(const ConvGemmParams& p);
size_t stem7x7_packed_elems();
void stem7x7_pack_filters(const float* g, int C, float* out);  // g: [64][C][7][7], C <= 4
int launch_stem7x7(const ConvGemmParams& p, void* stream);
// multi-problem (NetGroup): prepare_conv_multi / launch_conv_multi take kStreamHalf as a variant and end here; p.w must be the packed image
long stream1x1_prepare_multi(const ConvGemmParams& p, const ConvMultiTable& tb, int nprob);  // the grid, or -1
int launch_stream1x1_multi(const ConvMultiArgs& a, void* stream);                              // a.p.nprob, a.t as filled by the caller

// The remaining kernels take `esize` = bytes per device element (4 float / 2 _Float16); host-side tensors
// and the per-channel affine vectors are always float.

// MAX pooling, NHWC, windows clipped to the image (pooling_layer.cpp:140-187).
int launch_maxpool(const void* x, void* y, int esize, int NB, int H, int W, int C, int OH, int OW, int k, int s,
                   int pad, void* stream);

// y = act(x*a[c] + b[c] + z)   (a,b,z optional) — the stand-alone BatchNorm/Scale/ReLU/Eltwise/Sigmoid
// layers when they are not folded into a producing convolution.
int launch_eltwise(const void* x, const void* z, const float* a, const float* b, void* y, int esize, long total, int C,
                   int relu, int sigmoid, void* stream);

// crop the top-left (offset oh,ow) OH x OW window of an NHWC tensor (crop_layer.cpp:37-50)
int launch_crop(const void* x, void* y, int esize, int NB, int H, int W, int C, int oh, int ow, int OH, int OW,
                void* stream);

// layout changes at the Blob boundary (host side is NCHW float, blob.hpp:153-164)
// src NCHW [NB,C,H,W] -> dst NHWC with channel pitch CP (>= C, extra channels zeroed)
int launch_nchw_to_nhwc(const float* src, void* dst, int esize, int NB, int C, int H, int W, int CP, void* stream);
// src NHWC pitch CP, channels [c0, c0+C) -> dst NCHW [NB,C,H,W], float (dst_esize 4) or — from a half image only —
// _Float16 (dst_esize 2: the gather payload of an fp16 net)
int launch_nhwc_to_nchw(const void* src, void* dst, int esize, int NB, int C, int H, int W, int CP, int c0,
                        void* stream, int dst_esize = 4);
// packed filter image float -> half (fp16 nets)
int launch_f32_to_f16(const float* src, void* dst, long n, void* stream);

// pose decode (estimate_pose.py:131-143) from NHWC score / refinement maps (channel pitch + first channel)
// Multi-person consumers of the maps.  The reference repository stops at the maps; what these kernels
// invert is the label ENCODING of its training layer (src/caffe/layers/pose_data_layer.cpp:686-802): a cell (row, col)
// stands for the image point pt = (col*8 + 4, row*8 + 4) / scale; loc_pred holds (joint - pt)*scale / sqrt(53);
// next_pred channel pair l holds ((next joint - pt)*scale - mean[l]) / std[l] for regression edge l.
// part_select: per (image, joint) map, the local maxima (value >= thr, maximal in the (2r+1)^2 window, ties to the lower
//           cell index) ordered by (score desc, cell asc); the first max_det go to
//           out[((n*J+j)*max_det + k)*5 + {0..4}] = x, y, score, row, col (x, y refined with loc_pred and divided by scale),
//           counts[n*J+j] = how many were written.  Deterministic for every input (no arrival-order truncation).
//           spill: NB*J*H*W keys of scratch.
int launch_part_select(const void* prob, int pcp, int pc0, const void* loc, int lcp, int lc0, int esize, int NB, int H, int W, int J, float thr,
                       int radius, double scale, int max_det, unsigned long long* spill, int* counts, double* out, void* stream);
// pairwise: out[(d*E + l)*2 + k] = pt_k + (next_pred[2l+k] at the detection's cell * std[l][k] + mean[l][k]) / scale
int launch_pairwise_decode(const void* next, int ncp, int nc0, int esize, int NB, int H, int W, int E, double scale, int ndet,
                           const int* det /* [ndet][3] image, row, col */, const double* mean, const double* stdev, double* out,
                           void* stream);

// Image pre-processing of the demo (python/pose/estimate_pose.py:83-103) on the device: replicate padding by
// coordinate clamping, Pillow's two-pass 8-bit bilinear resample (22-bit fixed-point weights from the host),
// mean subtraction and the zero canvas, written straight into the network's NHWC input image.
struct ImagePrepParams {
  const unsigned char* src;  // [n][h][w][3] BGR uint8
  int n, h, w;
  int out_h, out_w;          // canvas = the network input
  int use_h, use_w;          // top-left part of the resized image that lands on the canvas; the rest is zero
  const int* x_bounds;       // [new_w][2] first tap, tap count; nullptr: no horizontal pass
  const int* x_coeffs;       // [new_w][x_ksize]
  int x_ksize;
  const int* y_bounds;       // nullptr: no vertical pass
  const int* y_coeffs;
  int y_ksize;
  unsigned char* tmp;        // [n][rows][use_w][4]: rows row0..row0+rows of the horizontally resampled padded image
  int row0, rows;
  void* dst;                 // [n][out_h][out_w][dst_cp] float (dst_esize 4) or _Float16 (2); pad channels zeroed
  int dst_esize, dst_cp;
  float mean[3];
};
int launch_image_prep(const ImagePrepParams& p, void* stream);

int launch_pose_decode(const void* prob, int pcp, int pc0, const void* loc, int lcp, int lc0, int esize, int NB, int H,
                       int W, int J, double scale, double* out, void* stream);

}  // namespace dc
