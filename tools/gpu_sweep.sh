#!/bin/bash
# ON THE GPU BOX: in-flight sweep — forwards in flight x forced tile variants (diagnostics for DESIGN 7b / 8b)
set -u
TAG=${1:-sweep}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cp profiles/r02_tune_cache.txt $OUT/tune_cache.txt
run() {  # label, env..., -- args
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
for s in 2 3 4 6; do run tuned_s$s DC_TUNE_CACHE=$OUT/tune_cache.txt -- --streams $s; done
for s in 3 6; do run nowino_s$s DC_WINOGRAD=0 -- --streams $s; done
for v in 9 5 3 10; do for s in 3 4 6; do run var${v}_s$s DC_CONV_VARIANT=$v -- --streams $s; done; done
