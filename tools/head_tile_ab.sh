#!/bin/bash
# ON THE GPU BOX: the merged 406-channel deconvolution head (float32, batch 1) forced to each multi-class tile in turn, everything
# else on its tuned tile: `value` (3 forwards in flight), one forward at a time, coalesced requests — and, with `pmc`, the
# launch's HBM-side bytes.   gpurun -- 'bash tools/head_tile_ab.sh <tag> [pmc]'
TAG=${1:-head}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export DC_TUNE_CACHE=$OUT/tune_cache.txt
rm -f $DC_TUNE_CACHE
timeout 400 python bench.py --no-cpu-baseline --no-f16-line --steps 60 --warmup 5 > $OUT/tuned.json 2> $OUT/err.txt
KEY=$(grep "+mc4" $DC_TUNE_CACHE | awk '{print $1}' | head -1); CUR=$(grep "+mc4" $DC_TUNE_CACHE | awk '{print $2}' | head -1)
echo "head key $KEY tuned to $CUR"
cp $DC_TUNE_CACHE $OUT/base_cache.txt
for v in $CUR 32x32x64_w114_p4 64x64x32_w221_p3 64x64x64_w222_p3 128x64x32_w222_p2 64x128x32_w222_p2 128x128x32_w222_p2; do
  grep -v "+mc4" $OUT/base_cache.txt > $DC_TUNE_CACHE; echo "$KEY $v" >> $DC_TUNE_CACHE
  for rep in 1 2; do
    timeout 400 python bench.py --no-cpu-baseline --no-f16-line --steps 60 --warmup 5 > $OUT/$v.json 2>> $OUT/err.txt
    python - "$v" $OUT/$v.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("%-22s value %.1f  one-at-a-time %.1f  coalesced %.1f" % (sys.argv[1], d["value"], d["one_forward_at_a_time"]["value"], d.get("cross_request_batching", {}).get("value", 0)))
PY
  done
  if [ "${2:-}" = pmc ]; then
    R=$PWD; cd /tmp; export TMPDIR=/tmp
    CMD="python $R/bench.py --no-cpu-baseline --no-f16-line --no-resnet101 --coalesce 0 --streams 1 --no-graph --steps 2 --warmup 1"
    timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/$OUT/f_$v -o f -- $CMD > /dev/null 2>> $R/$OUT/err.txt
    timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/$OUT/w_$v -o w -- $CMD > /dev/null 2>> $R/$OUT/err.txt
    cd $R
    python tools/pmc_per_shape.py $(find $OUT/f_$v -name "*.db" | head -1) $(find $OUT/w_$v -name "*.db" | head -1) | grep "multi-class" | head -2
    python tools/pmc_hbm_traffic.py $(find $OUT/f_$v -name "*.db" | head -1) $(find $OUT/w_$v -name "*.db" | head -1) "$v" 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('   overall', {k: d[k] for k in d if 'ratio' in k or 'per_launch' in k})" 2>/dev/null
    rm -rf $OUT/f_$v $OUT/w_$v
  fi
done
