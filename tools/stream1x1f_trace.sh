#!/bin/bash
# ON THE GPU BOX: rocprofv3 --kernel-trace --stats of 100 batch-1 float32 forwards with the conv4_x expansions on the tuned tile / on ws1x1f
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/wsf_trace; mkdir -p $OUT
for pin in tuned ws1x1f; do
  rm -rf /tmp/wsf_$pin; mkdir -p /tmp/wsf_$pin
  (R=$PWD; cd /tmp && DC_TUNE_CACHE=/tmp/wsf_tune.txt rocprofv3 --kernel-trace --stats -d /tmp/wsf_$pin -o t -- python $R/tools/stream1x1f_in_net.py --pin $pin > $OUT/$pin.log 2>&1)
  f=$(find /tmp/wsf_$pin -name "*_results.db" | head -1)
  echo "== $pin ($(grep forwards $OUT/$pin.log))"
  python tools/rocprof_summary.py $f | cut -c1-160 | head -16
done | tee $OUT/summary.txt
