# usage: bash tools/gpu_r6_lib_ab.sh <tag> <variant-name> [configs...]: same-box A/B of rainbow_amd/librainbow_hip_<variant>.so against the default library
TAG=$1; V=$2; shift; shift
mkdir -p gpurun_out
ROOT=$PWD
if [ -n "$PYTEST_K" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -q -k "$PYTEST_K" > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_gpu.log
  grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" gpurun_out/${TAG}_pytest_gpu.log | tail -3
fi
for cfg in "$@"; do
  CFG=$cfg ROUNDS=3 bash tools/gpu_env_ab.sh "RAINBOW_AMD_LIB=$ROOT/rainbow_amd/librainbow_hip_$V.so" "RAINBOW_AMD_LIB=$ROOT/rainbow_amd/librainbow_hip.so" 2>&1 | sed "s/^/$cfg /; s#$ROOT/rainbow_amd/##"
done | tee gpurun_out/${TAG}_ab.txt
bash tools/gpu_trace_gaps.sh $1 > gpurun_out/${TAG}_trace.txt 2>&1; grep "n/step" gpurun_out/${TAG}_trace.txt | cut -c1-110
