#!/bin/bash
# ON THE GPU BOX: A/B of two builds of the library (tools/probes/bin/libdeepcut_hip_old.so vs the in-tree one), interleaved
OUT=gpurun_out/${1:-oldnew}
mkdir -p $OUT
cp profiles/r02_tune_cache.txt $OUT/tune_cache.txt
for rep in 1 2 3; do for which in old new; do
  if [ $which = old ]; then export DEEPCUT_HIP_LIB=$PWD/tools/probes/bin/libdeepcut_hip_old.so; else unset DEEPCUT_HIP_LIB; fi
  DC_TUNE_CACHE=$OUT/tune_cache.txt timeout 200 python bench.py --no-cpu-baseline --no-f16-line --coalesce 0 --steps 150 --warmup 10 > $OUT/$which.json 2> $OUT/$which.err
  python - <<PY
import json
d=json.loads(open("$OUT/$which.json").read().strip().splitlines()[-1])
print("$which  value %.1f  one-at-a-time %.1f" % (d["value"], d["one_forward_at_a_time"]["value"]))
PY
done; done
