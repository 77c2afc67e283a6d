#!/bin/bash
# ON THE GPU BOX: rocprofv3 --kernel-trace --stats of 100 batch-1 float32 forwards (544x736) with the conv4_x expansions pinned on their best
# gather-GEMM tile / on ws1x1f: what a launch takes INSIDE the forward (cold filters, the previous kernel's tail), which a burst of
# identical launches does not show.  PINS="64x128x32_w222_p2 ws1x1f" REMAPS="1 0" bash tools/stream1x1f_trace.sh
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
R=$PWD; OUT=$PWD/gpurun_out/wsf_trace; mkdir -p $OUT
(cd /tmp && DC_TUNE_CACHE=/tmp/wsf_tune.txt python $R/tools/stream1x1f_in_net.py --pin tuned --forwards 2 > /dev/null 2>&1)   # the tuning launches stay out of the traces
for remap in ${REMAPS:-1}; do for pin in ${PINS:-64x128x32_w222_p2 ws1x1f}; do
  rm -rf /tmp/wsf_$pin; mkdir -p /tmp/wsf_$pin
  (cd /tmp && DC_WSF_REMAP=$remap DC_TUNE_CACHE=/tmp/wsf_tune.txt rocprofv3 --kernel-trace --stats -d /tmp/wsf_$pin -o t -- python $R/tools/stream1x1f_in_net.py --pin $pin > $OUT/$pin.log 2>&1)
  f=$(find /tmp/wsf_$pin -name "*_results.db" | head -1)
  echo "== conv4_x expansions on $pin, DC_WSF_REMAP=$remap ($(grep forwards $OUT/$pin.log))"
  python tools/rocprof_summary.py $f | cut -c1-160 | sed -n 2,9p; python tools/rocprof_summary.py $f | tail -2
done; done | tee $OUT/summary.txt
