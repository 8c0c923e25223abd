# round 6, second GPU call: the GPU suite (all of it), the folded conv slice reduction (RB_OPTS fold_reduce) as a same-box A/B on the three
# BASELINE configs, its trace, non-temporal load variants of the two HBM streams (variant builds of the same sources), and the trace of
# the opt-in early draw at batch 256 (slower than in round 5 on the first call's box: where?)
TAG=${1:-round6_second}
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$PWD
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_gpu.log
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" gpurun_out/${TAG}_pytest_gpu.log | tail -6
for cfg in pong-canonical-b32 breakout-canonical-b256 data-efficient-b32; do
  CFG=$cfg ROUNDS=3 bash tools/gpu_env_ab.sh "RB_OPTS=fold_reduce=0" "RB_OPTS=fold_reduce=1" 2>&1 | sed "s/^/$cfg /"
done | tee gpurun_out/${TAG}_fold_reduce_ab.txt
bash tools/gpu_trace_gaps.sh pong-canonical-b32 > gpurun_out/${TAG}_trace.txt 2>&1; grep "n/step" gpurun_out/${TAG}_trace.txt | cut -c1-110
for v in adamnt fwdnt; do
  if [ -f rainbow_amd/librainbow_hip_$v.so ]; then
    CFG=pong-canonical-b32 ROUNDS=3 bash tools/gpu_env_ab.sh "RAINBOW_AMD_LIB=$ROOT/rainbow_amd/librainbow_hip.so" "RAINBOW_AMD_LIB=$ROOT/rainbow_amd/librainbow_hip_$v.so" 2>&1 | sed "s/^/$v /"
  fi
done | tee gpurun_out/${TAG}_nt_ab.txt
RB_OPTS=spec_draw=1 bash tools/gpu_trace_gaps.sh breakout-canonical-b256 > gpurun_out/${TAG}_trace_b256_spec.txt 2>&1; grep "n/step" gpurun_out/${TAG}_trace_b256_spec.txt | cut -c1-110
bash tools/gpu_trace_gaps.sh breakout-canonical-b256 > gpurun_out/${TAG}_trace_b256.txt 2>&1; grep "n/step" gpurun_out/${TAG}_trace_b256.txt | cut -c1-110
timeout -k 10 120 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/${TAG}_20step_bench.json.log; cut -c1-200 gpurun_out/${TAG}_20step_bench.json.log
