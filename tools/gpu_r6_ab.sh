# usage: bash tools/gpu_r6_ab.sh <tag> <config> <rounds> "<RB_OPTS a>" "<RB_OPTS b>" [trace-opts]: GPU suite (quick subset optional), same-box A/B of two
# RB_OPTS settings on one config, then a kernel trace with the second setting
TAG=$1; CFG=$2; ROUNDS=$3; A=$4; B=$5
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ -n "$PYTEST_K" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -q -k "$PYTEST_K" > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_gpu.log
  grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" gpurun_out/${TAG}_pytest_gpu.log | tail -4
fi
CFG=$CFG ROUNDS=$ROUNDS bash tools/gpu_env_ab.sh "RB_OPTS=$A" "RB_OPTS=$B" 2>&1 | tee gpurun_out/${TAG}_ab.txt
RB_OPTS=$B bash tools/gpu_trace_gaps.sh $CFG > gpurun_out/${TAG}_trace.txt 2>&1; grep "n/step" gpurun_out/${TAG}_trace.txt | cut -c1-110
