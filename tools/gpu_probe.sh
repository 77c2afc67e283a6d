#!/bin/bash
# ON THE GPU BOX: tools/gpu_probe.sh <tag> -- the conv_probe A/B table (tools/probes/conv_probe.cpp); extra args go to the probe
OUT=gpurun_out/${1:-probe}; shift
mkdir -p $OUT
# correctness on the ragged shapes first, under a short timeout: a barrier bug must not hang the box
timeout 60 tools/probes/bin/conv_probe --shapes tiny_3x3,tiny_d2,tiny_odd,tiny_w1,tiny_w3 --batch 3 --reps 5 "$@" > $OUT/tiny.txt 2>&1 || { echo "tiny run failed/hung"; tail -20 $OUT/tiny.txt; exit 1; }
cat $OUT/tiny.txt
if grep -q WRONG $OUT/tiny.txt; then echo "WRONG results on the tiny shapes: not timing"; exit 1; fi
timeout 300 tools/probes/bin/conv_probe "$@" > $OUT/probe.txt 2>&1
cat $OUT/probe.txt
