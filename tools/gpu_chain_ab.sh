# usage: bash tools/gpu_chain_ab.sh [config] — same-box A/B of the chained conv launches (RB_CONV_CHAIN=0/1), alternating rounds
CFG=${1:-pong-canonical-b32}
mkdir -p gpurun_out
for r in 1 2 3; do
  for c in ${CHAIN_MODES:-0 1 2 3}; do
    RB_CONV_CHAIN=$c timeout 200 python bench.py --config $CFG --steps 1000 --warmup 200 --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('chain=$c round $r: %.2f us/step  %.0f steps/s' % (d['ms_per_step']*1e3, d['value']))"
  done
done
