#!/bin/bash
set -u
TAG=${1:-sweep2}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cp profiles/r02_tune_cache.txt $OUT/tune_cache.txt
run() {
  local label=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  env "${envs[@]}" timeout 200 python bench.py --no-cpu-baseline --no-f16-line --coalesce 0 --steps 30 --warmup 3 "$@" > $OUT/$label.json 2> $OUT/$label.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/$label.json").read().strip().splitlines()[-1])
    print("%-28s value %.1f  one-at-a-time %.1f" % ("$label", d["value"], d["one_forward_at_a_time"]["value"]))
except Exception as e:
    print("$label failed", e)
PY
}
for q in 2 4 8; do for s in 3 4 5 6 8; do run q${q}_s$s GPU_MAX_HW_QUEUES=$q DC_TUNE_CACHE=$OUT/tune_cache.txt -- --streams $s; done; done
for s in 4 6 8; do run q8_var9_s$s GPU_MAX_HW_QUEUES=8 DC_CONV_VARIANT=9 -- --streams $s; done
for s in 2 3 4; do run q8_b2_s$s GPU_MAX_HW_QUEUES=8 -- --streams $s --batch 2; done
