#!/bin/bash
# ON THE GPU BOX: MFMA counters of ONE executor's dispatches while other executors (another process, unprofiled) keep forwards in
# flight on the same GPU.  rocprofv3 serialises the dispatches of the process it profiles, not the neighbours': the SQ counters of a
# dispatch window count every wave on the chip, so this reads the matrix pipes' utilisation of the in-flight regime directly — and,
# since round 6, BY NETWORK STAGE (res3 / res4 / res5: raw SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE, the north-star's "conv3-conv5"),
# for the float32 batch-1 forward (three neighbours: `value`'s four in flight) and for the float16 batch-8 forward (one neighbour:
# configs[2]'s two in flight).
#   gpurun -- 'bash tools/pmc_in_flight.sh <tag> [f32|f16]'
set -u
TAG=${1:?tag}; WHAT=${2:-f32}; R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export DC_TUNE_CACHE=$OUT/tune_cache.txt
SRC=${TUNE:-$(ls profiles/r*_tune_cache.txt | sort | tail -1)}; [ "$SRC" != none ] && cp $SRC $DC_TUNE_CACHE
MF="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
if [ $WHAT = f16 ]; then ARGS="--dtype f16 --batch 8"; NEIGH="1 150 f16 8"; NDESC="ONE more executor of ANOTHER, unprofiled process keeps float16 batch-8 forwards in flight"
else ARGS=""; NEIGH="3 150"; NDESC="three more executors of ANOTHER, unprofiled process keep batch-1 forwards in flight"; fi
Q="--no-cpu-baseline --no-f16-line --no-resnet101 --coalesce 0"
# the plan (launch order -> stage) and the unprofiled duration of every launch, same tiles: a plain run first (it also warms the tune cache)
timeout 400 python bench.py $Q --streams 1 --steps 10 --warmup 2 $ARGS --breakdown $OUT/per_launch_$WHAT.txt > $OUT/bench_s1_$WHAT.json 2> $OUT/bench_s1.err
CMD="python $R/bench.py $Q --streams 1 --no-graph --steps 3 --warmup 1 $ARGS"
cd /tmp && export TMPDIR=/tmp
# (a) alone
timeout 300 rocprofv3 --kernel-trace --pmc $MF -d $OUT/alone -o m -- $CMD > /dev/null 2> $OUT/alone.err
# (b) with the neighbours in flight
rm -f /tmp/dc_load_ready
python $R/tools/background_load.py $NEIGH > $OUT/load_$WHAT.txt 2>&1 &
LOAD=$!
for i in $(seq 1 120); do [ -f /tmp/dc_load_ready ] && break; sleep 1; done
sleep 7   # the neighbours alone: their own rate is on record before the profiled pass starts
echo "profiled pass starts t=$(date +%s.%N)" >> $OUT/load_marks.txt
timeout 300 rocprofv3 --kernel-trace --pmc $MF -d $OUT/loaded -o m -- $CMD > /dev/null 2> $OUT/loaded.err
echo "profiled pass ends   t=$(date +%s.%N)" >> $OUT/load_marks.txt
kill $LOAD 2>/dev/null; wait $LOAD 2>/dev/null
cd $R
for w in alone loaded; do
  DB=$(find $OUT/$w -name "*.db" | head -1)
  python tools/pmc_mfma_util.py $DB "rocprofv3 --kernel-trace --pmc $MF over one executor (bench.py --streams 1 --no-graph --steps 3 $ARGS), $w$( [ $w = loaded ] && echo ": $NDESC on the same GPU")" $OUT/per_launch_$WHAT.txt > $OUT/pmc_mfma_util_${WHAT}_$w.txt 2> $OUT/post_$w.err
  grep -A8 "^stage" $OUT/pmc_mfma_util_${WHAT}_$w.txt
done
cat $OUT/load_marks.txt; grep -c . $OUT/load_$WHAT.txt; head -4 $OUT/load_$WHAT.txt; tail -3 $OUT/load_$WHAT.txt
rm -rf $OUT/alone $OUT/loaded
