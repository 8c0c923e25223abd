# round 5, first GPU call: the whole GPU suite on the head (new: Agent-default hosted pass at cfg 2/3/4, exchange at world 8,
# failed-draw marks, staged lazy operands), then the head's kernel trace and an unbracketed bench line as the round's baseline
TAG=${1:-r5a}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_gpu.log
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" gpurun_out/${TAG}_pytest_gpu.log | tail -30
bash tools/gpu_trace_gaps.sh pong-canonical-b32 > gpurun_out/${TAG}_trace.txt 2>&1; grep "n/step" gpurun_out/${TAG}_trace.txt | cut -c1-120
timeout 200 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | cut -c1-300
