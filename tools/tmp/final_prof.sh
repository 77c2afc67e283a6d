set -u
R=$PWD; OUT=$R/gpurun_out/r05finalprof; mkdir -p $OUT
export DC_TUNE_CACHE=$OUT/tune_cache.txt
cp profiles/r05_tune_cache.txt $DC_TUNE_CACHE
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o s -- python $R/bench.py --no-cpu-baseline --no-f16-line --no-resnet101 --coalesce 0 --streams 1 > $OUT/bench_under_rocprof.json 2> $OUT/rocprof_stats.err
cd $R
db=$(find $OUT/stats -name "*.db" | head -1)
python tools/rocprof_summary.py $db > $OUT/kernel_stats.txt 2> $OUT/post.err
python tools/rocprof_gaps.py $db > $OUT/kernel_gaps.txt 2>> $OUT/post.err
rm -rf $OUT/stats
tail -4 $OUT/kernel_stats.txt; head -3 $OUT/kernel_gaps.txt
