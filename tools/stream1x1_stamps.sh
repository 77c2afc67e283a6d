#!/bin/bash
# phase stamps of the forced streaming launch of one shape (tools/stream1x1_probe.py --stamps)
cd "$(dirname "$0")/.."
for sh in ${SHAPES:-res4c}; do
  DC_DEBUG_TIMING=0 python tools/stream1x1_probe.py --stamps --shapes $sh 2>&1 | grep -A1 "dc timing" | cut -c1-400
done
