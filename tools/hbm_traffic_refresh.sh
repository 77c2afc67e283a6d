R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06x; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
export DC_TUNE_CACHE=$OUT/tune_cache.txt; cp $R/profiles/r06_tune_cache.txt $DC_TUNE_CACHE
Q="--no-cpu-baseline --no-f16-line --no-resnet101 --coalesce 0 --streams 1 --no-graph"
PMC_CMD="python $R/bench.py $Q --steps 3 --warmup 1"
PMC16="python $R/bench.py $Q --dtype f16 --batch 8 --steps 2 --warmup 1"
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o f -- $PMC_CMD > /dev/null 2> $OUT/pmc_fetch.err
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o w -- $PMC_CMD > /dev/null 2> $OUT/pmc_write.err
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch16 -o f -- $PMC16 > /dev/null 2> $OUT/pmc_fetch16.err
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write16 -o w -- $PMC16 > /dev/null 2> $OUT/pmc_write16.err
cd $R
db() { find $OUT/$1 -name "*.db" | head -1; }
python tools/pmc_hbm_traffic.py $(db pmc_fetch) $(db pmc_write) "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over \`bench.py $Q --steps 3 --warmup 1\`, r06x (tiles of profiles/r06_tune_cache.txt; every convolution launch counted, ws1x1f included)" > $OUT/pmc_hbm_traffic.json
python tools/pmc_per_shape.py $(db pmc_fetch) $(db pmc_write) > $OUT/pmc_hbm_traffic_per_shape.txt
python tools/pmc_hbm_traffic.py $(db pmc_fetch16) $(db pmc_write16) "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over \`bench.py $Q --dtype f16 --batch 8 --steps 2 --warmup 1\`, r06x (every convolution launch counted, ws1x1 / stem7x7 included)" 8.4115e9 > $OUT/pmc_hbm_traffic_f16_b8.json
python tools/pmc_per_shape.py $(db pmc_fetch16) $(db pmc_write16) > $OUT/pmc_hbm_traffic_per_shape_f16_b8.txt
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_fetch16 $OUT/pmc_write16
grep -n "hbm_bytes_per_launch\|dispatches" $OUT/pmc_hbm_traffic.json $OUT/pmc_hbm_traffic_f16_b8.json | tail -20
