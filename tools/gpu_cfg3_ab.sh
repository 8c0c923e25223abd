# usage: bash tools/gpu_cfg3_ab.sh — config 3 (batch 256): default vs RB_GENERIC_FC=1, and the per-kernel trace of the default
mkdir -p gpurun_out
for r in 1 2; do
  for v in default generic_fc; do
    E=""; [ $v = generic_fc ] && E="RB_GENERIC_FC=1"
    env $E timeout 300 python bench.py --config breakout-canonical-b256 --steps 400 --warmup 100 --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v round $r: %.1f us/step  %.0f steps/s' % (d['ms_per_step']*1e3, d['value']))"
  done
done
bash tools/gpu_trace_gaps.sh breakout-canonical-b256 | grep "n/step"
