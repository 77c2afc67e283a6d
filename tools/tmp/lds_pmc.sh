set -u
R=$PWD; OUT=$R/gpurun_out/r05lds; mkdir -p $OUT
export DC_TUNE_CACHE=$OUT/tune_cache.txt; cp gpurun_out_seed_tune.txt $DC_TUNE_CACHE 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/pmc -o l -- python $R/bench.py --no-cpu-baseline --no-f16-line --no-resnet101 --coalesce 0 --streams 1 --no-graph --steps 3 --warmup 1 > /dev/null 2> $OUT/pmc.err
cd $R
python tools/pmc_lds_conflicts.py $(find $OUT/pmc -name "*.db" | head -1) > $OUT/lds_conflicts_f32_b1.txt 2>> $OUT/pmc.err; cat $OUT/lds_conflicts_f32_b1.txt
rm -rf $OUT/pmc
cd /tmp
timeout 400 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/pmc16 -o l -- python $R/bench.py --no-cpu-baseline --no-f16-line --no-resnet101 --coalesce 0 --dtype f16 --batch 8 --streams 1 --no-graph --steps 2 --warmup 1 > /dev/null 2> $OUT/pmc16.err
cd $R
python tools/pmc_lds_conflicts.py $(find $OUT/pmc16 -name "*.db" | head -1) > $OUT/lds_conflicts_f16_b8.txt 2>> $OUT/pmc16.err; cat $OUT/lds_conflicts_f16_b8.txt
rm -rf $OUT/pmc16
tail -3 $OUT/pmc.err
