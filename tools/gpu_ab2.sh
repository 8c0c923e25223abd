set -x
mkdir -p gpurun_out
rm -f gpurun_out/ab2.log
for rep in 1 2; do for O in 0 1; do
  RAINBOW_AMD_FUSED_UPDATE=$O timeout 300 python bench.py --steps 1500 --warmup 200 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused_update=$O', round(d['value'],1), 'steps/s', round(d['ms_per_step']*1000,1),'us')" >> gpurun_out/ab2.log
done; done
cat gpurun_out/ab2.log
