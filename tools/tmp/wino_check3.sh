set -u
R=$PWD; OUT=$R/gpurun_out/r05wino3; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_winograd.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
for idx in 42; do
DC_DEBUG_TIMING=$idx timeout 200 python bench.py --no-cpu-baseline --no-f16-line --no-resnet101 --coalesce 0 --no-graph --streams 1 --steps 2 --warmup 1 2>&1 >/dev/null | grep -A1 "dc timing" | tail -3
done
export DC_TUNE_CACHE=$OUT/tune_cache.txt; rm -f $DC_TUNE_CACHE
timeout 500 python bench.py --no-cpu-baseline --no-f16-line --no-resnet101 --coalesce 0 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
python - $OUT/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "one at a time", d["one_forward_at_a_time"]["value"], "frac", d["roofline"]["frac"], d["config"]["tile_tuning"])
PY
