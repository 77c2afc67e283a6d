#!/bin/bash
# ON THE GPU BOX: the default bench line (tiles tuned from scratch in every process) with the float32 streaming 1x1 form among the
# autotuner's candidates (DC_STREAM1X1 unset) and without it (=0), interleaved, fresh processes.   bash tools/stream1x1f_bench_ab.sh [reps]
cd "$(dirname "$0")/.."
OUT=gpurun_out/wsf_ab; mkdir -p $OUT
for rep in $(seq 1 ${1:-2}); do for mode in on off; do
  if [ $mode = off ]; then export DC_STREAM1X1=0; else unset DC_STREAM1X1; fi
  timeout 600 python bench.py --no-cpu-baseline --no-f16-line --no-resnet101 --coalesce 0 > $OUT/run_${mode}_$rep.json 2> $OUT/run_${mode}_$rep.err
  python - $OUT/run_${mode}_$rep.json $mode <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ws1x1f %-3s: value %.1f [%.1f-%.1f]  one forward at a time %.1f (frac %.4f)  in flight frac %.4f" % (
    sys.argv[2], d["value"], d["value_min"], d["value_max"], d["one_forward_at_a_time"]["value"], d["roofline"]["frac"], d["roofline"]["frac_in_flight"]))
PY
done; done | tee $OUT/summary.txt
