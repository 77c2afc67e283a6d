#!/bin/bash
# ON THE GPU BOX: tile tuning from scratch with and without the in-situ pass interleaved:
#   gpurun -- 'bash tools/insitu_ab.sh [f16|f32|both]'
mkdir -p gpurun_out/insitu
W=${1:-both}
run() {  # run <label> <bench args...>  (environment of the caller)
  local label=$1; shift
  rm -f gpurun_out/insitu/tc.txt
  DC_TUNE_CACHE=gpurun_out/insitu/tc.txt python bench.py --no-cpu-baseline --no-f16-line --no-resnet101 --coalesce 0 "$@" 2>>gpurun_out/insitu/err.txt |
    python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s value %7.1f  one at a time %7.1f' % ('$label', d['value'], d['one_forward_at_a_time']['value']))"
}
for rep in 1 2; do
  if [ $W != f32 ]; then
    DC_TUNE_INSITU=0 run "f16 b8  pass 1 only" --dtype f16 --batch 8 --streams 2 --steps 30 --warmup 4
    run "f16 b8  in situ 12%/4" --dtype f16 --batch 8 --streams 2 --steps 30 --warmup 4
  fi
  if [ $W != f16 ]; then
    DC_TUNE_INSITU=0 run "f32 b1  pass 1 only" --steps 60 --warmup 5
    run "f32 b1  in situ 12%/4" --steps 60 --warmup 5
  fi
done
