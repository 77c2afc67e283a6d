#!/bin/bash
# ON THE GPU BOX: P processes x S executors of batch-1 forwards on ONE GPU (tools/background_load.py), their rates summed — does a second
# process (its own four hardware queues) buy what a fifth stream of one process does not?  (diagnostic; the product is one process per GPU)
#   gpurun -- 'bash tools/multi_process_load.sh <tag> "1x4 2x2 2x3 2x4 3x2"'
set -u
TAG=${1:?tag}; SETS=${2:-"1x3 1x4 2x2 2x3 2x4"}; R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export DC_TUNE_CACHE=$OUT/tune_cache.txt
SRC=$(ls profiles/r*_tune_cache.txt | sort | tail -1); cp $SRC $DC_TUNE_CACHE
for set in $SETS; do
  P=${set%x*}; S=${set#*x}; pids=""
  for p in $(seq 1 $P); do
    DC_LOAD_READY=/tmp/dc_ready_$p python tools/background_load.py $S 22 > $OUT/load_${set}_$p.txt 2>&1 &
    pids="$pids $!"
  done
  for pid in $pids; do wait $pid; done
  # the steady lines (every process up): the last four 3-second windows of each process
  python - $OUT $set $P <<'PY'
import re, sys
out, st, P = sys.argv[1], sys.argv[2], int(sys.argv[3])
tot = 0.0
for p in range(1, P + 1):
    v = [float(m.group(1)) for m in re.finditer(r"t=[\d.]+\s+([\d.]+) images/s", open("%s/load_%s_%d.txt" % (out, st, p)).read())]
    tot += sum(v[-4:-1]) / max(1, len(v[-4:-1]))
print("%s (processes x executors): %.1f images/s in total" % (st, tot))
PY
done | tee $OUT/summary.txt
