#!/bin/bash
# ON THE GPU BOX: device-side phase stamps of f16 batch-8 launches
OUT=gpurun_out/${1:-timing16}; shift
mkdir -p $OUT
cp profiles/r02_tune_cache.txt $OUT/tune_cache.txt
for idx in "$@"; do
  DC_TUNE_CACHE=$OUT/tune_cache.txt DC_DEBUG_TIMING=$idx timeout 200 python bench.py --no-cpu-baseline --no-f16-line --coalesce 0 --no-graph --dtype f16 --batch 8 --streams 1 --steps 2 --warmup 1 2>&1 >/dev/null | grep -A1 "dc timing" | tail -3
done
DC_TUNE_CACHE=$OUT/tune_cache.txt timeout 200 python bench.py --no-cpu-baseline --no-f16-line --coalesce 0 --dtype f16 --batch 8 --streams 1 --steps 10 --warmup 2 --breakdown $OUT/per_launch.txt > $OUT/bench.json 2>$OUT/bench.err
python tools/breakdown.py $OUT/per_launch.txt | head -14
