# usage: bash tools/gpu_cfgs_ab.sh <libA.so> <libB.so> — same-box A/B of two builds of the library on the three bench configs
for cfg in pong-canonical-b32 data-efficient-b32 breakout-canonical-b256; do
  for r in 1 2; do
    for lib in "$@"; do
      RAINBOW_AMD_LIB=$PWD/$lib timeout 120 python bench.py --config $cfg --steps 1000 --warmup 200 --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read() or '{\"ms_per_step\":0,\"value\":0}'); print('[$cfg $lib] round $r: %.2f us/step  %.0f steps/s' % (d['ms_per_step']*1e3, d['value']))"
    done
  done
done
