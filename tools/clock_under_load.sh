#!/bin/bash
# ON THE GPU BOX: the shader clock the chip sustains idle, under ONE batch-1 forward at a time and under FOUR in flight
# (tools/probes/clock_probe from a process of its own beside tools/background_load.py).  Build the probe first (in the build container):
#   hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/probes/clock_probe.hip -o tools/probes/bin/clock_probe
#   gpurun -- 'bash tools/clock_under_load.sh <tag>'
set -u
TAG=${1:?tag}; R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export DC_TUNE_CACHE=$OUT/tune_cache.txt
SRC=$(ls profiles/r*_tune_cache.txt | sort | tail -1); cp $SRC $DC_TUNE_CACHE
P=tools/probes/bin/clock_probe
{
echo "# idle GPU"
$P 4 20 50
for SET in ${2:-"1:f32:1 4:f32:1 2:f16:8"}; do
  S=${SET%%:*}; DT=$(echo $SET | cut -d: -f2); B=${SET##*:}
  rm -f /tmp/dc_load_ready
  python tools/background_load.py $S 40 $DT $B > $OUT/load_$S.txt 2>&1 &
  LOAD=$!
  for i in $(seq 1 120); do [ -f /tmp/dc_load_ready ] && break; sleep 1; done
  sleep 4
  echo "# $S batch-$B forward(s) of the 544x736 $DT ResNet-152 in flight (another process)"
  $P 12 20 50
  kill $LOAD 2>/dev/null; wait $LOAD 2>/dev/null
  grep "images/s" $OUT/load_$S.txt | tail -2
done
} 2>&1 | tee $OUT/clock_under_load.txt
