# usage: bash tools/gpu_env_ab.sh "<VAR=a VAR2=b>" "<VAR=c>" ... — same-box A/B of environment settings on the headline config, 3 alternating rounds
mkdir -p gpurun_out
for r in $(seq 1 ${ROUNDS:-3}); do
  for v in "$@"; do
    env $v timeout 200 python bench.py --config ${CFG:-pong-canonical-b32} --steps 1000 --warmup 200 --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$v] round $r: %.2f us/step  %.0f steps/s' % (d['ms_per_step']*1e3, d['value']))"
  done
done
