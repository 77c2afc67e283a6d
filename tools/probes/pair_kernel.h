// pair_kernel.h — interface of the bottleneck-pair experiment (tools/probes/pair_kernel.hip, timed by pair_probe).
// NOT part of libdeepcut_hip.so: the fused form measured slower than the two gather-GEMM launches it would replace
// (EXPERIMENTS.md, round 5), so the product keeps the two launches.
#pragma once
#include <cstddef>

namespace dc {

// ---- bottleneck pair (float16): res<i>_branch2c + shortcut + ReLU and res<i+1>_branch2a + ReLU as ONE launch;
//      the 4*WD-wide intermediate is stored (the next block's shortcut) but never read back: it feeds the second GEMM from
//      registers.  Replaces two gather-GEMM launches (base_conv_layer.cpp:326-341 twice, eltwise_layer.cpp:59-65,
//      relu_layer.cpp:15-18) for 1x1 / stride-1 layers over dense NHWC tensors.
constexpr int kMaxPairProblems = 16;
struct PairProblem {
  const void* b;  // [M][WD]   float16: branch2b's output
  const void* s;  // [M][4*WD] float16: the block's input (shortcut)
  void* y;        // [M][4*WD] float16: the block's output
  void* z;        // [M][WD]   float16: the next block's branch2a output
  int M;
  int tile0;      // first workgroup of the problem (filled by pair_grid)
};
struct PairArgs {
  const void* w;     // pair_pack_filters() image
  const float* ab1;  // [4*WD] scale then [4*WD] shift of the first layer (folded BatchNorm + Scale, x the filter row scaling)
  const float* ab2;  // [WD] scale then [WD] shift of the second layer
  int WD;            // bottleneck width: 128 (conv3_x) or 256 (conv4_x)
  int relu2;
  int nprob;
  int pad_;
  long long* dbg;    // diagnostic build only: [workgroups][256] shader-clock stamps of wave 0 (else null)
  PairProblem prob[kMaxPairProblems];
};
bool pair_supported(int WD);
size_t pair_packed_halves(int WD);  // float16 elements of the filter image
size_t pair_lds_bytes(int WD);
// w1 [4*WD][WD], w2 [WD][4*WD] (K contiguous, float) -> the stage images the kernel streams (float16 bit patterns)
void pair_pack_filters(const float* w1, const float* w2, int WD, unsigned short* out);
long pair_grid(PairArgs& a);  // fills prob[].tile0, returns the workgroups
int launch_pair_gemm(const PairArgs& a, long grid, void* stream);
// the first layer alone in the same persistent form (Y only; reads the W2c stages of the same filter image)
int launch_p2c_gemm(const PairArgs& a, long grid, void* stream);

}  // namespace dc
