# round 5: the split optimiser pass (RB_OPTS adam_split) — parity first (GPU twins), then a same-box A/B on the headline config and
# the kernel trace with the split on
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_learner_gpu.py -q -k "deferred_update_agent or agent_default_flag_set" > gpurun_out/r5_split_pytest.log 2>&1; tail -5 gpurun_out/r5_split_pytest.log
ROUNDS=3 bash tools/gpu_env_ab.sh "RB_OPTS=adam_split=0" "RB_OPTS=adam_split=1" 2>&1 | tee gpurun_out/r5_split_ab.txt
RB_OPTS=adam_split=1 bash tools/gpu_trace_gaps.sh pong-canonical-b32 > gpurun_out/r5_split_trace.txt 2>&1; grep "n/step" gpurun_out/r5_split_trace.txt | cut -c1-120
