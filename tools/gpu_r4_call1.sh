# round 4, GPU call 1: GPU tests, same-box A/B of the T16 conv forward + narrow fc_z dX, kernel trace of the default
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/r4c1_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4c1_pytest.log
ROUNDS=2 bash tools/gpu_env_ab.sh "RB_OPTS=t16=0,z_narrow=0" "RB_OPTS=t16=6,z_narrow=0" "RB_OPTS=t16=6,z_narrow=1" "RB_OPTS=t16=7,z_narrow=1" 2>&1 | tee gpurun_out/r4c1_ab.txt
bash tools/gpu_trace_gaps.sh pong-canonical-b32 2>&1 | tee gpurun_out/r4c1_trace.txt | head -40
