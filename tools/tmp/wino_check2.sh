set -u
R=$PWD; OUT=$R/gpurun_out/r05wino2; mkdir -p $OUT
for rep in 1 2; do
export DC_TUNE_CACHE=$OUT/tune_cache$rep.txt; rm -f $DC_TUNE_CACHE
timeout 500 python bench.py --no-cpu-baseline --no-f16-line --no-resnet101 --coalesce 0 --steps 20 --warmup 5 --breakdown $OUT/per_launch$rep.txt > $OUT/bench$rep.json 2> $OUT/bench$rep.err
python - $OUT/bench$rep.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "one at a time", d["one_forward_at_a_time"]["value"], "frac", d["roofline"]["frac"], d["config"]["tile_tuning"])
PY
grep "+w" $DC_TUNE_CACHE
done
python tools/breakdown.py $OUT/per_launch2.txt > $OUT/per_shape_summary.txt; head -8 $OUT/per_shape_summary.txt
