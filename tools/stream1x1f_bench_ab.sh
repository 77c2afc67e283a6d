mkdir -p gpurun_out/wsf8
for mode in -1 0 -1 0; do
  DC_STREAM1X1=$mode timeout 600 python bench.py --no-cpu-baseline --no-f16-line --no-resnet101 --coalesce 0 > gpurun_out/wsf8/run_$mode.json 2> gpurun_out/wsf8/run_$mode.err
  python - gpurun_out/wsf8/run_$mode.json $mode <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("DC_STREAM1X1=%s: value %.1f [%.1f-%.1f]  one at a time %.1f frac %.4f" % (sys.argv[2], d["value"], d["value_min"], d["value_max"], d["one_forward_at_a_time"]["value"], d["roofline"]["frac"]))
PY
done
