#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): produces under gpurun_out/$1/ everything profiles/ is refreshed from, already
# summarised as text.      gpurun --timeout 1500 -- 'bash tools/refresh_profiles.sh r02p'   then   bash tools/collect_profiles.sh r02p r02
# 1. the default bench line (tuning included) + the tune cache it ends with
# 2. one forward at a time with the per-launch hipEvent table; float16 batch 8
# 3. rocprofv3 --kernel-trace --stats over the one-forward-at-a-time bench (warm cache)
# 4. PMC passes, each on its own: MFMA utilisation counters, FETCH_SIZE, WRITE_SIZE (--kernel-trace only)
# 5. the other BASELINE configs end to end (tools/run_configs.py)
# 6. the probes (tools/probes/*): build them first: tools/probes/build_conv_probe.sh; hipcc ... l2_lds_probe.hip / bw_probe.hip
set -u
TAG=${1:-r02p}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
# 0. the line exactly as the driver asks for it (fresh tuning, no cache), with the wall time of the whole command
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_style.json 2> $OUT/bench_driver_style.err ) 2> $OUT/bench_driver_style.time
export DC_TUNE_CACHE=$OUT/tune_cache.txt
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 300 python bench.py --no-cpu-baseline --no-f16-line --no-resnet101 --coalesce 0 --streams 1 --breakdown $OUT/per_launch.txt > $OUT/bench_s1.json 2>> $OUT/bench.err
python tools/breakdown.py $OUT/per_launch.txt > $OUT/per_shape_summary.txt 2>> $OUT/bench.err
timeout 300 python bench.py --no-cpu-baseline --no-f16-line --no-resnet101 --coalesce 0 --dtype f16 --batch 8 --streams 2 --steps 20 --warmup 3 --breakdown $OUT/per_launch_f16_b8.txt > $OUT/bench_f16_batch8.json 2>> $OUT/bench.err
python tools/breakdown.py $OUT/per_launch_f16_b8.txt > $OUT/per_shape_summary_f16_b8.txt 2>> $OUT/bench.err
timeout 300 python bench.py --no-cpu-baseline --no-f16-line --no-resnet101 --coalesce 0 --config 3 --steps 10 --warmup 2 > $OUT/bench_config3_f32.json 2>> $OUT/bench.err
timeout 300 python bench.py --no-cpu-baseline --no-f16-line --no-resnet101 --coalesce 0 --config 3 --dtype f16 --steps 10 --warmup 2 > $OUT/bench_config3_f16.json 2>> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o s -- python $R/bench.py --no-cpu-baseline --no-f16-line --no-resnet101 --coalesce 0 --streams 1 > $OUT/bench_under_rocprof.json 2> $OUT/rocprof_stats.err
PMC_CMD="python $R/bench.py --no-cpu-baseline --no-f16-line --no-resnet101 --coalesce 0 --streams 1 --no-graph --steps 3 --warmup 1"
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/pmc_mfma -o m -- $PMC_CMD > /dev/null 2> $OUT/pmc_mfma.err
MF="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
# the same counters on the batched forwards: batch 2 (what cross-request batching runs) and batch 8 (configs[3]'s share)
for B in 2 8; do
  timeout 300 python $R/bench.py --no-cpu-baseline --no-f16-line --no-resnet101 --coalesce 0 --streams 1 --batch $B --steps 5 --warmup 2 --breakdown $OUT/per_launch_b$B.txt > $OUT/bench_s1_b$B.json 2>> $OUT/bench.err
  timeout 400 rocprofv3 --kernel-trace --pmc $MF -d $OUT/pmc_mfma_b$B -o m -- python $R/bench.py --no-cpu-baseline --no-f16-line --no-resnet101 --coalesce 0 --streams 1 --batch $B --no-graph --steps 2 --warmup 1 > /dev/null 2> $OUT/pmc_mfma_b$B.err
done
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o f -- $PMC_CMD > /dev/null 2> $OUT/pmc_fetch.err
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o w -- $PMC_CMD > /dev/null 2> $OUT/pmc_write.err
timeout 400 rocprofv3 --kernel-trace --pmc $MF -d $OUT/pmc_mfma16 -o m -- python $R/bench.py --no-cpu-baseline --no-f16-line --no-resnet101 --coalesce 0 --dtype f16 --batch 8 --streams 1 --no-graph --steps 2 --warmup 1 > /dev/null 2> $OUT/pmc_mfma16.err
# HBM traffic of the float16 batch-8 forward (configs[2]'s unit): minimum = 8 x 2.07 GB / 2 of activations + 0.263 GB / 2 of filters
PMC16="python $R/bench.py --no-cpu-baseline --no-f16-line --no-resnet101 --coalesce 0 --dtype f16 --batch 8 --streams 1 --no-graph --steps 2 --warmup 1"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch16 -o f -- $PMC16 > /dev/null 2> $OUT/pmc_fetch16.err
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write16 -o w -- $PMC16 > /dev/null 2> $OUT/pmc_write16.err
cd $R
db() { find $OUT/$1 -name "*.db" | head -1; }
python tools/rocprof_summary.py $(db stats) > $OUT/kernel_stats.txt 2> $OUT/post.err
python tools/rocprof_gaps.py $(db stats) > $OUT/kernel_gaps.txt 2>> $OUT/post.err
python tools/pmc_mfma_util.py $(db pmc_mfma) "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE over \`bench.py --no-cpu-baseline --no-f16-line --no-resnet101 --coalesce 0 --streams 1 --no-graph --steps 3 --warmup 1\` with a warm DC_TUNE_CACHE (tools/refresh_profiles.sh): one forward at a time" $OUT/per_launch.txt $OUT/bench.json > $OUT/pmc_mfma_util.txt 2>> $OUT/post.err
for B in 2 8; do
  python tools/pmc_mfma_util.py $(db pmc_mfma_b$B) "the same counters over \`bench.py --no-cpu-baseline --no-f16-line --no-resnet101 --coalesce 0 --streams 1 --batch $B --no-graph --steps 2 --warmup 1\` (float32, batch $B, one forward at a time)" $OUT/per_launch_b$B.txt $OUT/bench_s1_b$B.json > $OUT/pmc_mfma_util_batch$B.txt 2>> $OUT/post.err
done
python tools/pmc_hbm_traffic.py $(db pmc_fetch) $(db pmc_write) "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over \`bench.py --no-cpu-baseline --no-f16-line --no-resnet101 --coalesce 0 --streams 1 --no-graph --steps 3 --warmup 1\`, $TAG" > $OUT/pmc_hbm_traffic.json 2>> $OUT/post.err
python tools/pmc_per_shape.py $(db pmc_fetch) $(db pmc_write) > $OUT/pmc_hbm_traffic_per_shape.txt 2>> $OUT/post.err
python tools/pmc_mfma_util.py $(db pmc_mfma16) "the same counters over \`bench.py --no-cpu-baseline --no-f16-line --no-resnet101 --coalesce 0 --dtype f16 --batch 8 --streams 1 --no-graph --steps 2 --warmup 1\` (float16 operands, batch 8, one forward at a time; v_mfma_f32_32x32x16_f16 = 32 busy cycles)" $OUT/per_launch_f16_b8.txt $OUT/bench_f16_batch8.json > $OUT/pmc_mfma_util_f16_b8.txt 2>> $OUT/post.err
python tools/pmc_hbm_traffic.py $(db pmc_fetch16) $(db pmc_write16) "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over \`bench.py --no-cpu-baseline --no-f16-line --no-resnet101 --coalesce 0 --dtype f16 --batch 8 --streams 1 --no-graph --steps 2 --warmup 1\`, $TAG" 8.4115e9 > $OUT/pmc_hbm_traffic_f16_b8.json 2>> $OUT/post.err
python tools/pmc_per_shape.py $(db pmc_fetch16) $(db pmc_write16) > $OUT/pmc_hbm_traffic_per_shape_f16_b8.txt 2>> $OUT/post.err
timeout 600 python tools/run_configs.py > $OUT/configs.json 2> $OUT/configs.err
# 6. the RCCL process group with one rank; the stand-alone probes only with PROBES=1 (the single-problem kernels they time did not
#    change in round 4: profiles/r03_conv_probe_*.txt, r03_l2_lds_probe.txt, r03_bw_probe.txt stand)
if [ "${PROBES:-0}" = 1 ]; then
  timeout 300 tools/probes/bin/conv_probe --stamps > $OUT/conv_probe_f16_b8.txt 2>&1
  timeout 300 tools/probes/bin/conv_probe --dtype f --batch 1 --reps 50 > $OUT/conv_probe_f32_b1.txt 2>&1
  timeout 120 tools/probes/bin/l2_lds_probe > $OUT/l2_lds_probe.txt 2>&1
  timeout 120 tools/probes/bin/bw_probe > $OUT/bw_probe.txt 2>&1
fi
timeout 120 python tools/rccl_world1_check.py > $OUT/rccl_world1_check.txt 2>&1
# 7. the grouped float16 pyramid (round 4): per-launch table, MFMA counters, HBM traffic, kernel stats and gaps
unset DC_TUNE_CACHE
bash tools/gpu_group_pmc.sh $TAG > $OUT/group_pmc.log 2>&1
rm -rf $OUT/stats $OUT/pmc_mfma $OUT/pmc_mfma_b2 $OUT/pmc_mfma_b8 $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_fetch16 $OUT/pmc_write16 $OUT/pmc_mfma16
ls -la $OUT | head -60
tail -c 400 $OUT/bench.json; tail -5 $OUT/post.err
