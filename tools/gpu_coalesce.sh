#!/bin/bash
# ON THE GPU BOX: cross-request batching sweep (coalesce k x executors)
OUT=gpurun_out/${1:-coal}
mkdir -p $OUT
cp profiles/r02_tune_cache.txt $OUT/tune_cache.txt
for s in 1 2 3; do for c in 2 3 4 6 8; do
  DC_TUNE_CACHE=$OUT/tune_cache.txt timeout 300 python bench.py --no-cpu-baseline --no-f16-line --coalesce $c --streams $s --steps 40 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); x=d['cross_request_batching']
print('executors $s coalesce $c: %.1f images/s   (value %.1f)' % (x['value'], d['value']))"
done; done
