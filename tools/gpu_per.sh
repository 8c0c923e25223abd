# usage: bash tools/gpu_per.sh — the PER-only path: GPU parity of the replay kernels, then tools/per_bench.py with lazy priority
# write-backs (update + sample as one launch) against immediate ones, and the headline bench (the sampler body was refactored)
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_replay_gpu.py tests/test_abi.py -q -x 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -3
for v in "RAINBOW_AMD_LAZY_PRIORITIES=0" "RAINBOW_AMD_LAZY_PRIORITIES=1" "RAINBOW_AMD_LAZY_PRIORITIES=0" "RAINBOW_AMD_LAZY_PRIORITIES=1"; do
  echo "[$v]"; env $v timeout 120 python tools/per_bench.py 2>/dev/null | tail -1
done
echo "[B=256]"; SAMPLE_CONFIG=breakout-canonical-b256 timeout 120 python tools/per_bench.py 2>/dev/null | tail -1
timeout 120 python bench.py --steps 1000 --warmup 200 --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | cut -c1-200
