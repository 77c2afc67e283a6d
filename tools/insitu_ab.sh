mkdir -p gpurun_out/i1
for rep in 1 2; do for v in 0 1; do
  rm -f gpurun_out/i1/tc_$v.txt
  DC_TUNE_INSITU=$v DC_TUNE_CACHE=gpurun_out/i1/tc_$v.txt python bench.py --no-cpu-baseline --no-f16-line --coalesce 0 --dtype f16 --batch 8 --streams 2 --steps 30 --warmup 4 2>gpurun_out/i1/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f16 insitu=$v', round(d['value'],1), round(d['one_forward_at_a_time']['value'],1))"
done; done
for rep in 1 2; do for v in 0 1; do
  rm -f gpurun_out/i1/tc32_$v.txt
  DC_TUNE_INSITU=$v DC_TUNE_CACHE=gpurun_out/i1/tc32_$v.txt python bench.py --no-cpu-baseline --no-f16-line --coalesce 0 --steps 60 --warmup 5 2>>gpurun_out/i1/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f32 insitu=$v', round(d['value'],1), round(d['one_forward_at_a_time']['value'],1))"
done; done
diff gpurun_out/i1/tc_0.txt gpurun_out/i1/tc_1.txt | head -20
