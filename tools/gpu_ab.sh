# usage: bash tools/gpu_ab.sh ENVVAR [bench args]  — same-box A/B of a 0/1 environment switch (two alternating repeats)
set -x
VAR=$1; shift
mkdir -p gpurun_out
rm -f gpurun_out/ab_$VAR.log
for rep in 1 2; do for O in 0 1; do
  env $VAR=$O timeout 300 python bench.py --steps 1500 --warmup 200 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$O', round(d['value'],1), 'steps/s', round(d['ms_per_step']*1000,1),'us')" >> gpurun_out/ab_$VAR.log
done; done
cat gpurun_out/ab_$VAR.log
