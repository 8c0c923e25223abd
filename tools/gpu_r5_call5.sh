# round 5: evidence refresh after the PMC tag fix and the `donate` option of update_priorities: cfg 2 / 3 / 4 bench + kernel stats + PMC, PER bench
bash tools/gpu_evidence.sh round5_final_cfg2 pong-canonical-b32
bash tools/gpu_evidence.sh round5_final_cfg4 data-efficient-b32
bash tools/gpu_evidence.sh round5_final_cfg3 breakout-canonical-b256
for v in "RAINBOW_AMD_LAZY_PRIORITIES=0" "RAINBOW_AMD_LAZY_PRIORITIES=1 PER_DONATE=0" "RAINBOW_AMD_LAZY_PRIORITIES=1 PER_DONATE=1"; do echo "[$v]"; env $v timeout 120 python tools/per_bench.py 2>/dev/null | tail -1; done > gpurun_out/round5_final_per_bench.txt
cat gpurun_out/round5_final_per_bench.txt
timeout 120 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/round5_final_20step_bench.json.log
