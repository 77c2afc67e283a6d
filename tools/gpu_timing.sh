#!/bin/bash
# ON THE GPU BOX: device-side phase stamps (DC_DEBUG_TIMING) of the given launch indices
OUT=gpurun_out/${1:-timing}; shift
mkdir -p $OUT
cp profiles/r02_tune_cache.txt $OUT/tune_cache.txt
for idx in "$@"; do
  DC_TUNE_CACHE=$OUT/tune_cache.txt DC_DEBUG_TIMING=$idx timeout 200 python bench.py --no-cpu-baseline --no-f16-line --coalesce 0 --no-graph --streams 1 --steps 2 --warmup 1 2>&1 >/dev/null | grep -A1 "dc timing" | tail -3
done
