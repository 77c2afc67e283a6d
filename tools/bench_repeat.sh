#!/bin/bash
# ON THE GPU BOX: the default bench line N times in fresh processes (tiles re-tuned, streams re-chosen by every run): what the spread
# between runs is, and whether the library's stream choice depends on the process.   gpurun -- 'bash tools/bench_repeat.sh <tag> [N]'
TAG=${1:?tag}; N=${2:-3}
OUT=gpurun_out/$TAG; mkdir -p $OUT
for i in $(seq 1 $N); do
  timeout 600 python bench.py --no-cpu-baseline --no-f16-line --no-resnet101 --coalesce 0 > $OUT/run$i.json 2> $OUT/run$i.err
done
python - $OUT $N <<'PY' | tee $OUT/bench_repeat.txt
import json, sys
out, n = sys.argv[1], int(sys.argv[2])
print("# bench.py --no-cpu-baseline --no-f16-line --no-resnet101 --coalesce 0, %d fresh processes on one box" % n)
for i in range(1, n + 1):
    try:
        d = json.loads(open("%s/run%d.json" % (out, i)).read().strip().splitlines()[-1])
        c = d["config"]["stream_choice"]
        print("run %d: value %.1f [%.1f-%.1f]  one at a time %.1f  host pipelined %.1f | streams: chosen %.1f, first created %.1f (ratio %.3f)"
              % (i, d["value"], d["value_min"], d["value_max"], d["one_forward_at_a_time"]["value"], d.get("pcie_inclusive_pipelined", {}).get("value", 0),
                 c["images_per_s_chosen"], c["images_per_s_first_created"], c["images_per_s_first_created"] / c["images_per_s_chosen"]))
    except Exception as e:  # noqa: BLE001
        print("run %d failed: %s" % (i, e))
PY
