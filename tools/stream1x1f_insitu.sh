#!/bin/bash
# ON THE GPU BOX: tools/stream1x1f_insitu.py for both forms of one conv4_x expansion, warm (three isolated launches) and in situ
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/wsf_insitu
IDX=$(DC_AUTOTUNE=0 python tools/stream1x1f_insitu.py --find 2>/dev/null | tail -1)
echo "launch index $IDX"
for insitu in 0 1; do for mode in 0 1; do
  echo "== DC_STREAM1X1=$mode DC_DEBUG_TIMING_INSITU=$insitu"
  DC_DEBUG_TIMING=$IDX DC_DEBUG_TIMING_INSITU=$insitu DC_AUTOTUNE=0 DC_STREAM1X1=$mode timeout 300 python tools/stream1x1f_insitu.py 2>&1 | grep -A1 "dc timing" | cut -c1-400
done; done | tee gpurun_out/wsf_insitu/stamps.log
