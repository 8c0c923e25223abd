# usage: bash tools/gpu_fc_gemm_tl.sh — timeline of the GEMM launches (RB_STAMP build) + A/B + kernel durations
mkdir -p gpurun_out
if [ -f rainbow_amd/librainbow_hip_stamp.so ]; then
  RAINBOW_AMD_LIB=$PWD/rainbow_amd/librainbow_hip_stamp.so timeout 120 python tools/stamp/gemm_timeline.py 2>&1 | grep "==\|   " | tee gpurun_out/gemm_timeline.txt
fi
for r in 1 2; do
  for v in ${AB:-"RB_OPTS=fc_gemm=0" "RB_OPTS=fc_gemm=-1"}; do
    env $v timeout 90 python bench.py --config breakout-canonical-b256 --steps 1000 --warmup 200 --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read() or '{\"ms_per_step\":0,\"value\":0}'); print('[$v] round $r: %.2f us/step  %.0f steps/s' % (d['ms_per_step']*1e3, d['value']))"
  done
done
bash tools/gpu_trace_gaps.sh breakout-canonical-b256 > gpurun_out/fc_gemm_trace_b256.txt 2>&1; grep "n/step" gpurun_out/fc_gemm_trace_b256.txt | grep "gemm" | cut -c1-130
