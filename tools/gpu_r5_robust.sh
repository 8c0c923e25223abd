# round 5: robustness of the cross-stream pieces — the GPU suite with every launch serialised and blocking (the in-kernel waits must
# terminate when launches run strictly in submission order), then the soak run in the shape of main.py's loop
mkdir -p gpurun_out
AMD_SERIALIZE_KERNEL=3 HIP_LAUNCH_BLOCKING=1 timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/round5_final_serialized_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/round5_final_serialized_pytest_gpu.log
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" gpurun_out/round5_final_serialized_pytest_gpu.log | tail -4
SOAK_STEPS=40000 timeout 600 python tools/soak.py > gpurun_out/round5_final_soak.json.log 2>&1; tail -1 gpurun_out/round5_final_soak.json.log | cut -c1-400
