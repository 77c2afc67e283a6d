set -u
TAG=r05w; R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_style.json 2> $OUT/bench_driver_style.err ) 2> $OUT/bench_driver_style.time
export DC_TUNE_CACHE=$OUT/tune_cache.txt
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
cp $DC_TUNE_CACHE $OUT/tune_cache_after_bench.txt
# one forward at a time: latency-tuned tiles (a cache of its own: the in-flight descent of the run above re-tiled the shared one)
export DC_TUNE_CACHE=$OUT/tune_cache_latency.txt
timeout 300 python bench.py --no-cpu-baseline --no-f16-line --no-resnet101 --coalesce 0 --streams 1 --breakdown $OUT/per_launch.txt > $OUT/bench_s1.json 2>> $OUT/bench.err
python tools/breakdown.py $OUT/per_launch.txt > $OUT/per_shape_summary.txt 2>> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o s -- python $R/bench.py --no-cpu-baseline --no-f16-line --no-resnet101 --coalesce 0 --streams 1 > $OUT/bench_under_rocprof.json 2> $OUT/rocprof_stats.err
PMC_CMD="python $R/bench.py --no-cpu-baseline --no-f16-line --no-resnet101 --coalesce 0 --streams 1 --no-graph --steps 3 --warmup 1"
MF="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
timeout 400 rocprofv3 --kernel-trace --pmc $MF -d $OUT/pmc_mfma -o m -- $PMC_CMD > /dev/null 2> $OUT/pmc_mfma.err
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o f -- $PMC_CMD > /dev/null 2> $OUT/pmc_fetch.err
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o w -- $PMC_CMD > /dev/null 2> $OUT/pmc_write.err
timeout 400 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/pmc_lds -o l -- $PMC_CMD > /dev/null 2> $OUT/pmc_lds.err
cd $R
db() { find $OUT/$1 -name "*.db" | head -1; }
python tools/rocprof_summary.py $(db stats) > $OUT/kernel_stats.txt 2> $OUT/post.err
python tools/rocprof_gaps.py $(db stats) > $OUT/kernel_gaps.txt 2>> $OUT/post.err
python tools/pmc_mfma_util.py $(db pmc_mfma) "rocprofv3 --kernel-trace --pmc $MF over \`bench.py --no-cpu-baseline --no-f16-line --no-resnet101 --coalesce 0 --streams 1 --no-graph --steps 3 --warmup 1\` with a warm DC_TUNE_CACHE: one forward at a time (end of round 5: Winograd kernel with the conflict-free LDS layout, buffer addressing, 16-wave form on res4)" $OUT/per_launch.txt $OUT/bench_s1.json > $OUT/pmc_mfma_util.txt 2>> $OUT/post.err
python tools/pmc_hbm_traffic.py $(db pmc_fetch) $(db pmc_write) "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over \`bench.py --no-cpu-baseline --no-f16-line --no-resnet101 --coalesce 0 --streams 1 --no-graph --steps 3 --warmup 1\`, $TAG" > $OUT/pmc_hbm_traffic.json 2>> $OUT/post.err
python tools/pmc_per_shape.py $(db pmc_fetch) $(db pmc_write) > $OUT/pmc_hbm_traffic_per_shape.txt 2>> $OUT/post.err
python tools/pmc_lds_conflicts.py $(db pmc_lds) > $OUT/pmc_lds_conflicts.txt 2>> $OUT/post.err
rm -rf $OUT/stats $OUT/pmc_mfma $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_lds
tail -4 $OUT/kernel_stats.txt; head -6 $OUT/pmc_lds_conflicts.txt; cat $OUT/pmc_hbm_traffic.json | head -c 600; tail -3 $OUT/post.err
python - <<'PY'
import json
for f in ("bench_driver_style","bench"):
    d=json.loads(open('gpurun_out/r05w/%s.json'%f).read().strip().splitlines()[-1])
    print(f, "value", d["value"], "one at a time", d["one_forward_at_a_time"]["value"], "frac", d["roofline"]["frac"], d["config"]["tile_tuning"][:120])
PY
