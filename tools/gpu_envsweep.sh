# usage: bash tools/gpu_envsweep.sh VAR "v1 v2 ..." [bench args] — same-box sweep of one environment knob (two alternating passes)
VAR=$1; VALS=$2; shift; shift
mkdir -p gpurun_out
rm -f gpurun_out/sweep_$VAR.log
for rep in 1 2; do for V in $VALS; do
  env $VAR=$V timeout 300 python bench.py --steps 1500 --warmup 200 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$VAR=$V', round(d['value'],1), 'steps/s', round(d['ms_per_step']*1000,1),'us;', r['kernel'], round(r['avg_us'],2), 'us')" >> gpurun_out/sweep_$VAR.log
done; done
cat gpurun_out/sweep_$VAR.log
