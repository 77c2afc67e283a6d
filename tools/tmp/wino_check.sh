set -u
R=$PWD; OUT=$R/gpurun_out/r05wino; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_winograd.py tests/test_gpu_tuning.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -6 $OUT/pytest.log
export DC_TUNE_CACHE=$OUT/tune_cache.txt; rm -f $DC_TUNE_CACHE
timeout 500 python bench.py --no-cpu-baseline --no-f16-line --no-resnet101 --coalesce 0 --steps 20 --warmup 5 --breakdown $OUT/per_launch.txt > $OUT/bench.json 2> $OUT/bench.err
python tools/breakdown.py $OUT/per_launch.txt > $OUT/per_shape_summary.txt; head -12 $OUT/per_shape_summary.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05wino/bench.json').read().strip().splitlines()[-1])
print("value", d["value"], "one at a time", d["one_forward_at_a_time"]["value"], "frac", d["roofline"]["frac"], d["config"]["tile_tuning"])
PY
grep wino $OUT/tune_cache.txt | head
