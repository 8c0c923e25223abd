# usage: bash tools/gpu_fc_gemm.sh — the hidden layer's tiled GEMMs (fc_gemm.h) at batch 256: parity at the BASELINE shapes, same-box A/B
# against the streamed kernels (RB_OPTS fc_gemm=0), per-kernel trace.  Every step under its own short timeout.
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_learner_gpu.py -q -x -k "baseline_shapes or class_level" 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -4
for r in 1 2; do
  for v in "RB_OPTS=fc_gemm=0" "RB_OPTS=fc_gemm=-1"; do
    env $v timeout 90 python bench.py --config breakout-canonical-b256 --steps 1000 --warmup 200 --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read() or '{\"ms_per_step\":0,\"value\":0}'); print('[$v] round $r: %.2f us/step  %.0f steps/s' % (d['ms_per_step']*1e3, d['value']))"
  done
done
sed -i 's/timeout 600 rocprofv3/timeout 150 rocprofv3/' tools/gpu_trace_gaps.sh
bash tools/gpu_trace_gaps.sh breakout-canonical-b256 > gpurun_out/fc_gemm_trace_b256.txt 2>&1; grep "n/step" gpurun_out/fc_gemm_trace_b256.txt | cut -c1-130
