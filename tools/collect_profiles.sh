#!/bin/bash
# HERE (build container): copy the summaries a tools/refresh_profiles.sh run left in gpurun_out/<tag>/ into profiles/<prefix>_*
TAG=${1:-r02p}; PFX=${2:-r02}
S=gpurun_out/$TAG
for f in bench.json bench_driver_style.json bench_driver_style.time bench_f16_batch8.json bench_under_rocprof.json bench_config3_f32.json bench_config3_f16.json configs.json kernel_stats.txt kernel_gaps.txt per_shape_summary.txt per_shape_summary_f16_b8.txt pmc_mfma_util.txt pmc_mfma_util_batch2.txt pmc_mfma_util_batch8.txt pmc_mfma_util_f16_b8.txt pmc_hbm_traffic.json pmc_hbm_traffic_per_shape.txt pmc_hbm_traffic_f16_b8.json pmc_hbm_traffic_per_shape_f16_b8.txt tune_cache.txt conv_probe_f16_b8.txt conv_probe_f32_b1.txt l2_lds_probe.txt bw_probe.txt rccl_world1_check.txt per_shape_summary_group.txt pmc_mfma_util_group.txt pmc_hbm_traffic_group.json pmc_hbm_traffic_per_shape_group.txt kernel_stats_group.txt kernel_gaps_group.txt group_timing.txt; do
  [ -s $S/$f ] && cp $S/$f profiles/${PFX}_$f
done
[ -s $S/per_launch.txt ] && cp $S/per_launch.txt profiles/${PFX}_per_launch_hipevents.txt
[ -s $S/group_per_launch.txt ] && cp $S/group_per_launch.txt profiles/${PFX}_per_launch_hipevents_group.txt
[ -s $S/group_per_launch_lanes2.txt ] && cp $S/group_per_launch_lanes2.txt profiles/${PFX}_per_launch_hipevents_group_lanes2.txt
[ -s $S/tune_cache_group.txt ] && cp $S/tune_cache_group.txt profiles/${PFX}_tune_cache_group.txt
ls profiles/${PFX}_*
