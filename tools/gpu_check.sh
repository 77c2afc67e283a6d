#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the -m gpu suite, the default bench line and a per-launch table.
#   gpurun --timeout 1500 -- 'bash tools/gpu_check.sh r02a'
set -u
TAG=${1:-r02a}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export DC_TUNE_CACHE=$OUT/tune_cache.txt
timeout 1100 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 1500 $OUT/bench.json
timeout 300 python bench.py --no-cpu-baseline --no-f16-line --streams 1 --breakdown $OUT/per_launch.txt > $OUT/bench_s1.json 2>> $OUT/bench.err
tail -3 $OUT/bench.err
