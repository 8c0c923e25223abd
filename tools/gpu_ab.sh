# A/B of launch strategies on the same box
set -x
mkdir -p gpurun_out
for G in 0 1; do for S in 0 1; do
  RAINBOW_AMD_GRAPH=$G RB_NO_SIDE_STREAMS=$S timeout 300 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('graph=$G no_side=$S', round(d['value'],1), 'steps/s', round(d['ms_per_step']*1000,1),'us')" >> gpurun_out/ab.log
done; done
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 >> gpurun_out/ab.log
cat gpurun_out/ab.log
