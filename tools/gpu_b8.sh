#!/bin/bash
# ON THE GPU BOX: float32 batch-N forward, tiles tuned from scratch; per-shape table
OUT=gpurun_out/${1:-b8}; B=${2:-8}
mkdir -p $OUT
rm -f $OUT/tune_cache.txt
DC_TUNE_CACHE=$OUT/tune_cache.txt timeout 400 python bench.py --no-cpu-baseline --no-f16-line --coalesce 0 --batch $B --streams 2 --steps 10 --warmup 2 --breakdown $OUT/per_launch.txt > $OUT/bench.json 2>$OUT/bench.err
python - <<PY
import json
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("f32 batch $B: value %.1f img/s (%.0f TF/s)  one-at-a-time %.1f img/s (%.0f TF/s)" % (d["value"], d["tflops"], d["one_forward_at_a_time"]["value"], d["one_forward_at_a_time"]["tflops"]))
PY
python tools/breakdown.py $OUT/per_launch.txt | head -18
