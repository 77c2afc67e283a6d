#!/bin/bash
# ON THE GPU BOX: HBM-side traffic per kernel shape of the f16 batch-8 forward (FETCH_SIZE / WRITE_SIZE passes)
set -u
TAG=${1:-pmc16}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cp profiles/r02_tune_cache.txt $OUT/tune_cache.txt
export DC_TUNE_CACHE=$OUT/tune_cache.txt
cd /tmp && export TMPDIR=/tmp
PMC_CMD="python $R/bench.py --no-cpu-baseline --no-f16-line --coalesce 0 --dtype f16 --batch 8 --streams 1 --no-graph --steps 2 --warmup 1"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o f -- $PMC_CMD > /dev/null 2> $OUT/pmc_fetch.err
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o w -- $PMC_CMD > /dev/null 2> $OUT/pmc_write.err
cd $R
python tools/pmc_per_shape.py $(find $OUT/pmc_fetch -name "*.db" | head -1) $(find $OUT/pmc_write -name "*.db" | head -1) | head -24
