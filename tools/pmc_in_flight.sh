#!/bin/bash
# ON THE GPU BOX: MFMA counters of ONE executor's dispatches while THREE other executors (another process, unprofiled) keep forwards in
# flight on the same GPU.  rocprofv3 serialises the dispatches of the process it profiles, not the neighbours': if the SQ counters of a
# dispatch window count every wave on the chip, this reads the matrix pipes' utilisation of the in-flight regime directly.
#   gpurun -- 'bash tools/pmc_in_flight.sh <tag>'
set -u
TAG=${1:?tag}; R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export DC_TUNE_CACHE=$OUT/tune_cache.txt
SRC=$(ls profiles/r*_tune_cache.txt | sort | tail -1); cp $SRC $DC_TUNE_CACHE
MF="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
CMD="python $R/bench.py --no-cpu-baseline --no-f16-line --no-resnet101 --coalesce 0 --streams 1 --no-graph --steps 3 --warmup 1"
cd /tmp && export TMPDIR=/tmp
# (a) alone
timeout 300 rocprofv3 --kernel-trace --pmc $MF -d $OUT/alone -o m -- $CMD > /dev/null 2> $OUT/alone.err
# (b) with three neighbours in flight
rm -f /tmp/dc_load_ready
python $R/tools/background_load.py 3 150 > $OUT/load.txt 2>&1 &
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
  python tools/pmc_mfma_util.py $DB "rocprofv3 --kernel-trace --pmc $MF over one executor (bench.py --streams 1 --no-graph --steps 3), $w$( [ $w = loaded ] && echo ': three more executors of ANOTHER, unprofiled process keep batch-1 forwards in flight on the same GPU')" > $OUT/pmc_mfma_util_$w.txt 2> $OUT/post_$w.err
  tail -12 $OUT/pmc_mfma_util_$w.txt
done
cat $OUT/load_marks.txt; grep -c . $OUT/load.txt; head -4 $OUT/load.txt; tail -3 $OUT/load.txt
rm -rf $OUT/alone $OUT/loaded
