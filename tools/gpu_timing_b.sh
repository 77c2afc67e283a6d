#!/bin/bash
# ON THE GPU BOX: phase stamps of launches at a given batch (float32): gpu_timing_b.sh <tag> <batch> idx...
OUT=gpurun_out/${1:-timingb}; B=$2; shift; shift
mkdir -p $OUT; export DC_TUNE_CACHE=$OUT/tune_cache.txt
for idx in "$@"; do
  DC_DEBUG_TIMING=$idx timeout 200 python bench.py --no-cpu-baseline --no-f16-line --coalesce 0 --no-graph --batch $B --streams 1 --steps 2 --warmup 1 2>&1 >/dev/null | grep -A1 "dc timing" | tail -3
done
