# round 5: the early draw (RB_OPTS spec_draw) — GPU twins first, then same-box A/Bs on the three BASELINE configs, then traces
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_learner_gpu.py tests/test_replay_gpu.py -q -k "deferred_update_agent or gives_up or earlier_valid or lazy" > gpurun_out/r5_spec_pytest.log 2>&1; tail -4 gpurun_out/r5_spec_pytest.log
for cfg in pong-canonical-b32 data-efficient-b32 breakout-canonical-b256; do
  CFG=$cfg ROUNDS=2 bash tools/gpu_env_ab.sh "RB_OPTS=spec_draw=0" "RB_OPTS=spec_draw=1" 2>&1 | sed "s/^/$cfg /"
done | tee gpurun_out/r5_spec_ab.txt
for cfg in pong-canonical-b32 data-efficient-b32; do
  RB_OPTS=spec_draw=1 bash tools/gpu_trace_gaps.sh $cfg > gpurun_out/r5_spec_trace_$cfg.txt 2>&1; grep "n/step" gpurun_out/r5_spec_trace_$cfg.txt | cut -c1-120
done
