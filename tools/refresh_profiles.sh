#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): produces under gpurun_out/$1/ everything profiles/ is refreshed from.
#   gpurun --timeout 1500 -- 'bash tools/refresh_profiles.sh r02a'
# 1. the default bench line (tuning included) + the tune cache it ends with
# 2. rocprofv3 --kernel-trace --stats over the one-forward-at-a-time bench (warm cache)
# 3. PMC passes, each on its own: MFMA utilisation counters, FETCH_SIZE, WRITE_SIZE (--kernel-trace only)
set -u
TAG=${1:-r02a}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export DC_TUNE_CACHE=$OUT/tune_cache.txt
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 300 python bench.py --no-cpu-baseline --no-f16-line --streams 1 --breakdown $OUT/per_launch.txt > $OUT/bench_s1.json 2>> $OUT/bench.err
timeout 300 python bench.py --no-cpu-baseline --dtype f16 --batch 8 --streams 2 --steps 20 --warmup 3 > $OUT/bench_f16_b8.json 2>> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o s -- python $R/bench.py --no-cpu-baseline --no-f16-line --streams 1 > $OUT/bench_under_rocprof.json 2> $OUT/rocprof_stats.err
PMC_CMD="python $R/bench.py --no-cpu-baseline --no-f16-line --streams 1 --no-graph --steps 3 --warmup 1"
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/pmc_mfma -o m -- $PMC_CMD > /dev/null 2> $OUT/pmc_mfma.err
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o f -- $PMC_CMD > /dev/null 2> $OUT/pmc_fetch.err
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o w -- $PMC_CMD > /dev/null 2> $OUT/pmc_write.err
cd $R
ls -la $OUT $OUT/*/ | head -40
tail -c 600 $OUT/bench.json
