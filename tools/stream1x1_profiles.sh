#!/bin/bash
# round 6: the evidence files of the two float16 kernels added after the Winograd work, on the GPU box:
#   gpurun -- 'bash tools/stream1x1_profiles.sh gpurun_out/ws_prof'   ->  stream1x1_probe.txt, stem_probe.txt in that directory
set -u
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/ws_prof}; mkdir -p $OUT
F=$OUT/stream1x1_probe.txt
{
echo "# tools/stream1x1_probe.py on MI355X: the streaming form of the dense float16 1x1 layers (csrc/stream1x1.hip, ws1x1) against the gather-GEMM"
echo "# tiles on the 1x1 expansion layers of the 544x736 batch-8 forward.  Per shape: the forced streaming launch against a float64 evaluation of"
echo "# the float16 operands and against the forced direct launch; then the autotuner's isolated timings (5 launches back to back, best of 2 bursts)."
timeout 300 python tools/stream1x1_probe.py 2>&1 | grep -v amdgpu.ids
echo
echo "# the forms a K was measured on (DC_WS_ALT=<n>: the n-th form of its K; 0 = the one in use)"
for alt in 1 2; do echo "## DC_WS_ALT=$alt (K=128: four waves of two fragments; K=256: 1 = 512-channel slices of two fragments per wave, 2 = four waves, two workgroups per CU)"
  DC_WS_ALT=$alt timeout 200 python tools/stream1x1_probe.py --no-check --shapes res4c,res3c 2>&1 | grep -v amdgpu.ids; done
echo
echo "# phase stamps of the forced streaming launch (DC_DEBUG_TIMING; stream1x1 slots: prologue requests issued | first stage + filters landed |"
echo "# the peeled first D steps | the other steps + the late half's last epilogue | drain).  Cycles of s_memtime: comparable within a kernel only"
SHAPES="res4c res3c res5c" timeout 200 bash tools/stream1x1_stamps.sh
echo
echo "# for comparison: the stamps of the tile the autotuner took before (launch 61 = res4b7_branch2c of the batch-8 forward, d128x64x64_w221_s2, DC_STREAM1X1=0)"
DC_STREAM1X1=0 bash tools/gpu.sh stamps ws_prof_tmp --dtype f16 --batch 8 -- 61 2>&1 | cut -c1-400
} > $F 2>&1
{
echo "# tools/stem_probe.py on MI355X: conv1 (7x7/2, 3->64, +BN/Scale/ReLU) of a float16 net at 544x736, the autotuner's isolated timings of the stem"
echo "# kernel (csrc/stem_f16.hip, stem7x7) and of the row-tap gather-GEMM tiles"
timeout 200 python tools/stem_probe.py 2>&1 | grep -v amdgpu.ids
} > $OUT/stem_probe.txt 2>&1
tail -3 $F; cat $OUT/stem_probe.txt | tail -2
