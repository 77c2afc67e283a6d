#!/bin/bash
# ON THE GPU BOX: effect of runtime environment switches on the bench line
OUT=gpurun_out/${1:-envs}
mkdir -p $OUT
cp profiles/r02_tune_cache.txt $OUT/tune_cache.txt
run() {
  local label=$1; shift
  env DC_TUNE_CACHE=$OUT/tune_cache.txt "$@" timeout 200 python bench.py --no-cpu-baseline --no-f16-line --coalesce 0 --steps 40 --warmup 5 > $OUT/$label.json 2> $OUT/$label.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/$label.json").read().strip().splitlines()[-1])
    print("%-28s value %.1f  one-at-a-time %.1f" % ("$label", d["value"], d["one_forward_at_a_time"]["value"]))
except Exception as e:
    print("$label failed", e)
PY
}
run default A=1
run devkernarg1 HIP_FORCE_DEV_KERNARG=1
run devkernarg0 HIP_FORCE_DEV_KERNARG=0
run default2 A=1
