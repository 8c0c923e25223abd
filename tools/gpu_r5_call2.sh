# round 5, GPU call 2: the whole GPU suite on the head, then the SQ counter pass per BASELINE config (MFMA-busy, wave-cycle split,
# LDS conflicts of the round-4/5 kernels: k_conv_fwd_t16, k_fc_gemm_fwd/bwd, k_conv_dx_lds, k_act_fused)
TAG=${1:-r5b}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_gpu.log
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" gpurun_out/${TAG}_pytest_gpu.log | tail -12
for c in pong-canonical-b32 breakout-canonical-b256 data-efficient-b32; do bash tools/gpu_sqpmc.sh $c $TAG; done
