#!/bin/bash
# A/B timings of the float32 streaming 1x1 kernel (csrc/stream1x1_f32.hip) on the conv4_x expansion at batch 1: the autotuner's isolated
# burst timing of the forced form under DC_WSF_ABL (timing ablations, wrong results) and DC_WSF_REMAP.   ABLS="0 2" REMAPS="1 0" bash tools/wsf_ablate.sh
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/wsf_abl
for rep in 1 2; do
for remap in ${REMAPS:-1}; do
for abl in ${ABLS:-0 1 2 3 4 5}; do
  echo -n "rep $rep DC_WSF_REMAP=$remap DC_WSF_ABL=$abl: "
  DC_WSF_REMAP=$remap DC_WSF_ABL=$abl timeout 200 python tools/stream1x1_probe.py --dtype f32 --batch 1 --shapes ${SHAPE:-res4c} --no-check 2>&1 | grep chosen | sed 's/.*| best direct/best direct/'
done; done; done | tee gpurun_out/wsf_abl/abl.log
