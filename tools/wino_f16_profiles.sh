#!/bin/bash
# ON THE GPU BOX: everything profiles/rNN_wino_f16_*.txt is made from (round 6, DESIGN 4.1e / EXPERIMENTS I).  Build the ablation libraries
# and the two probes HERE first (bash tools/wino_f16_ablate.sh build; hipcc tools/probes/{mfma16,wino16_loop}_probe.hip -o tools/probes/bin/...).
#   gpurun --timeout 1500 -- 'bash tools/wino_f16_profiles.sh <tag>'
set -u
TAG=${1:?tag}; R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
python tools/wino_f16_probe.py 2>&1 | grep -v amdgpu.ids > $OUT/wino_f16_probe.txt
DC_DEBUG_TIMING=0 python tools/wino_f16_probe.py --stamps 2>&1 | grep -v amdgpu.ids | grep -A1 "^res\|launch 0" | grep -v "^--" >> $OUT/wino_f16_probe.txt
bash tools/wino_f16_ablate.sh run > $OUT/wino_f16_ablations.txt 2>&1
bash tools/wino_f16_pmc.sh > $OUT/wino_f16_pmc.txt 2>&1
tools/probes/bin/mfma16_probe > $OUT/mfma16_probe.txt 2>&1
tools/probes/bin/wino16_loop_probe > $OUT/wino16_loop_probe.txt 2>&1
# rocprofv3 --kernel-trace --stats of the float16 batch-8 bench, one forward at a time (the kernel's average duration inside the forward)
export DC_TUNE_CACHE=$OUT/tune_cache.txt
timeout 300 python bench.py --no-cpu-baseline --no-f16-line --no-resnet101 --coalesce 0 --dtype f16 --batch 8 --streams 1 --steps 10 --warmup 2 > $OUT/bench_f16_b8_s1.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats16 -o s -- python $R/bench.py --no-cpu-baseline --no-f16-line --no-resnet101 --coalesce 0 --dtype f16 --batch 8 --streams 1 --steps 20 --warmup 3 > $OUT/bench_f16_b8_under_rocprof.json 2> $OUT/rocprof16.err
cd $R
python tools/rocprof_summary.py $(find $OUT/stats16 -name "*.db" | head -1) > $OUT/kernel_stats_f16_b8.txt 2> $OUT/post.err
rm -rf $OUT/stats16
head -12 $OUT/kernel_stats_f16_b8.txt; cat $OUT/wino_f16_probe.txt | cut -c1-200 | head -12
