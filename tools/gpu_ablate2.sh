set -x
mkdir -p gpurun_out
for A in 0 0; do
  RB_ABLATE=$A timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ablate $A fc_h_fwd us', round(d['roofline']['avg_us'],2), 'step us', round(d['ms_per_step']*1000,1))"
done
