#!/bin/bash
# ON THE GPU BOX: the float16 4-scale pyramid as ONE grouped launch sequence under rocprofv3:
#   bash tools/gpu_group_pmc.sh <tag>      -> gpurun_out/<tag>/{group_per_launch.txt, per_shape_summary_group.txt, pmc_mfma_util_group.txt,
#                                             pmc_hbm_traffic_group.json, pmc_hbm_traffic_per_shape_group.txt, kernel_stats_group.txt}
# Counters are collected in their own runs (--kernel-trace + --pmc only), each pass separately, tiles from a cache a plain run wrote.
set -u
TAG=${1:?tag}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export DC_TUNE_CACHE=$OUT/tune_cache_group.txt
rm -f $DC_TUNE_CACHE
# timings: the default plan (two lanes of two scales, concurrent) and, for the per-launch / per-stage tables, ONE lane (every layer
# one launch over the four scales: launches do not overlap, so per-launch durations and counters mean something)
echo "# default lanes (two lanes of two scales each, concurrent on two streams)" > $OUT/group_timing.txt
timeout 600 python tools/group_profile.py --out $OUT/lanes2 --no-members --inflight 2 2>> $OUT/group.err | grep "^grouped\|^scale\|in flight\|merges" >> $OUT/group_timing.txt
cp $OUT/lanes2/group_per_launch.txt $OUT/group_per_launch_lanes2.txt
export DC_GROUP_LANES=1
rm -f $DC_TUNE_CACHE
echo "# DC_GROUP_LANES=1 (one lane: every layer ONE launch over the four scales)" >> $OUT/group_timing.txt
timeout 600 python tools/group_profile.py --out $OUT --no-members --inflight 2 2>> $OUT/group.err | grep "^grouped\|^scale\|in flight\|merges" >> $OUT/group_timing.txt
python tools/breakdown.py $OUT/group_per_launch.txt > $OUT/per_shape_summary_group.txt
cd /tmp && export TMPDIR=/tmp
MF="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
CMD="python $R/tools/group_profile.py --out $OUT/scratch --no-members --no-graph --pmc-run 2"
timeout 400 rocprofv3 --kernel-trace --pmc $MF -d $OUT/pmc_mfma -o m -- $CMD > /dev/null 2> $OUT/pmc_mfma.err
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o f -- $CMD > /dev/null 2> $OUT/pmc_fetch.err
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o w -- $CMD > /dev/null 2> $OUT/pmc_write.err
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/stats -o s -- python $R/tools/group_profile.py --out $OUT/scratch --no-members --pmc-run 20 > /dev/null 2> $OUT/stats.err
cd $R
db() { find $OUT/$1 -name "*.db" | head -1; }
python tools/pmc_mfma_util.py $(db pmc_mfma) "rocprofv3 --kernel-trace --pmc $MF over \`tools/group_profile.py --no-graph --pmc-run 2\`: the float16 4-scale pyramid of BASELINE configs[2] (batch 8 per scale) as ONE grouped launch sequence (caffe.NetGroup with DC_GROUP_LANES=1: every layer one multi-problem launch over the four scales), tiles from a warm cache; v_mfma_f32_32x32x16_f16 = 32 busy cycles" $OUT/group_per_launch.txt > $OUT/pmc_mfma_util_group.txt 2> $OUT/post.err
python tools/pmc_hbm_traffic.py $(db pmc_fetch) $(db pmc_write) "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over the grouped float16 pyramid (tools/group_profile.py --no-graph --pmc-run 2), $TAG" 28.08e9 > $OUT/pmc_hbm_traffic_group.json 2>> $OUT/post.err
python tools/pmc_per_shape.py $(db pmc_fetch) $(db pmc_write) > $OUT/pmc_hbm_traffic_per_shape_group.txt 2>> $OUT/post.err
python tools/rocprof_summary.py $(db stats) > $OUT/kernel_stats_group.txt 2>> $OUT/post.err
python tools/rocprof_gaps.py $(db stats) > $OUT/kernel_gaps_group.txt 2>> $OUT/post.err
rm -rf $OUT/pmc_mfma $OUT/pmc_fetch $OUT/pmc_write $OUT/stats $OUT/scratch $OUT/lanes2
unset DC_GROUP_LANES
cat $OUT/group_timing.txt | grep -v "^/opt"; tail -12 $OUT/pmc_mfma_util_group.txt; head -12 $OUT/pmc_hbm_traffic_group.json; tail -5 $OUT/post.err
