#!/bin/bash
# ON THE GPU BOX: interleaved comparison of several builds of the library (tools/probes/bin/lib_<name>.so; "base" = the in-tree
# one) with the SAME tile choices (profiles/r02_tune_cache.txt).   gpu_libs.sh <tag> name...
OUT=gpurun_out/${1:-libs}; shift
mkdir -p $OUT
for rep in 1 2; do for which in base "$@"; do
  if [ $which = base ]; then unset DEEPCUT_HIP_LIB; else export DEEPCUT_HIP_LIB=$PWD/tools/probes/bin/lib_$which.so; fi
  cp profiles/r02_tune_cache.txt $OUT/tune_cache.txt
  DC_TUNE_CACHE=$OUT/tune_cache.txt timeout 200 python bench.py --no-cpu-baseline --no-f16-line --coalesce 0 --steps 100 --warmup 10 > $OUT/$which.json 2> $OUT/$which.err
  DC_TUNE_CACHE=$OUT/tune_cache.txt timeout 200 python bench.py --no-cpu-baseline --no-f16-line --coalesce 0 --dtype f16 --batch 8 --steps 20 --warmup 3 > $OUT/${which}_f16.json 2>> $OUT/$which.err
  python - <<PY
import json
d=json.loads(open("$OUT/$which.json").read().strip().splitlines()[-1])
h=json.loads(open("$OUT/${which}_f16.json").read().strip().splitlines()[-1])
print("%-6s f32 b1: value %.1f one-at-a-time %.1f | f16 b8: %.1f / %.1f" % ("$which", d["value"], d["one_forward_at_a_time"]["value"], h["value"], h["one_forward_at_a_time"]["value"]))
PY
done; done
