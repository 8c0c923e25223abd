# usage: bash tools/gpu_sigma.sh — RB_LEARNER_IMPLICIT_SIGMA: Agent-level GPU tests, then a same-box A/B on the headline config
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_learner_gpu.py -q -x 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -4
ROUNDS=3 bash tools/gpu_env_ab.sh "RAINBOW_AMD_IMPLICIT_SIGMA=0" "RAINBOW_AMD_IMPLICIT_SIGMA=1"
RAINBOW_AMD_IMPLICIT_SIGMA=1 bash tools/gpu_trace_gaps.sh pong-canonical-b32 > gpurun_out/sigma_trace.txt 2>&1; grep "n/step" gpurun_out/sigma_trace.txt | cut -c1-120
