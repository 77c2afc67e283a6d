#!/bin/bash
# ON THE GPU BOX: quick A/B of a kernel change — bench lines (one at a time + 3 in flight) with a warm tune cache, a per-launch
# table, and the two HBM-traffic PMC passes.   gpurun --timeout 900 -- 'bash tools/gpu_ab.sh <tag>'
set -u
TAG=${1:-ab}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cp profiles/r02_tune_cache.txt $OUT/tune_cache.txt
export DC_TUNE_CACHE=$OUT/tune_cache.txt
timeout 300 python bench.py --no-cpu-baseline --no-f16-line --coalesce 0 --breakdown $OUT/per_launch.txt > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("value %.1f  one-at-a-time %.1f  frac %.3f" % (d["value"], d["one_forward_at_a_time"]["value"], d["roofline"]["frac"]))
PY
if [ "${2:-}" = "pmc" ]; then
cd /tmp && export TMPDIR=/tmp
PMC_CMD="python $R/bench.py --no-cpu-baseline --no-f16-line --coalesce 0 --streams 1 --no-graph --steps 3 --warmup 1"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o f -- $PMC_CMD > /dev/null 2> $OUT/pmc_fetch.err
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o w -- $PMC_CMD > /dev/null 2> $OUT/pmc_write.err
cd $R
python tools/pmc_hbm_traffic.py $(find $OUT/pmc_fetch -name "*.db" | head -1) $(find $OUT/pmc_write -name "*.db" | head -1) "$TAG" > $OUT/pmc_hbm_traffic.json 2> $OUT/pmc_traffic.err
cat $OUT/pmc_hbm_traffic.json | head -30
fi
