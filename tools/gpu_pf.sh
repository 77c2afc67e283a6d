#!/bin/bash
# ON THE GPU BOX: A/B of an environment switch (default DC_PREFETCH) with the same tune cache, interleaved repetitions
OUT=gpurun_out/${1:-pf}; VAR=${2:-DC_PREFETCH}
mkdir -p $OUT
cp profiles/r02_tune_cache.txt $OUT/tune_cache.txt
for rep in 1 2 3; do for pf in 0 1; do
  env $VAR=$pf DC_TUNE_CACHE=$OUT/tune_cache.txt timeout 200 python bench.py --no-cpu-baseline --no-f16-line --coalesce 0 --steps 150 --warmup 10 > $OUT/pf$pf.json 2> $OUT/pf$pf.err
  python - <<PY
import json
d=json.loads(open("$OUT/pf$pf.json").read().strip().splitlines()[-1])
print("$VAR=$pf  value %.1f  one-at-a-time %.1f" % (d["value"], d["one_forward_at_a_time"]["value"]))
PY
done; done
