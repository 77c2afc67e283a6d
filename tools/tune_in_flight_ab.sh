mkdir -p gpurun_out/tif
for rep in 1 2; do
for mode in "--no-tune-in-flight" ""; do
  rm -f gpurun_out/tif/tc.txt
  DC_TUNE_CACHE=gpurun_out/tif/tc.txt python bench.py --no-cpu-baseline --no-f16-line --no-resnet101 --coalesce 0 --steps 60 --warmup 5 $mode 2>>gpurun_out/tif/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f32 [%s]' % '$mode', round(d['value'],1), round(d['one_forward_at_a_time']['value'],1), d['config']['tile_tuning'])"
  rm -f gpurun_out/tif/tc.txt
  DC_TUNE_CACHE=gpurun_out/tif/tc.txt python bench.py --no-cpu-baseline --no-f16-line --no-resnet101 --coalesce 0 --dtype f16 --batch 8 --streams 2 --steps 30 --warmup 4 $mode 2>>gpurun_out/tif/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f16 [%s]' % '$mode', round(d['value'],1), round(d['one_forward_at_a_time']['value'],1), d['config']['tile_tuning'])"
done; done
tail -5 gpurun_out/tif/err.txt
