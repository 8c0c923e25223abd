# usage: bash tools/gpu_roundend.sh <tag>: the evidence set committed under profiles/ at the end of a round —
# GPU test log, smoke, and per bench config: bench line (with cpu_baseline), rocprofv3 kernel stats, PMC traffic; the loop bench
TAG=$1
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_gpu.log; tail -2 gpurun_out/${TAG}_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/${TAG}_smoke.log; tail -2 gpurun_out/${TAG}_smoke.log
bash tools/gpu_evidence.sh ${TAG}_cfg2 pong-canonical-b32
bash tools/gpu_evidence.sh ${TAG}_cfg3 breakout-canonical-b256
bash tools/gpu_evidence.sh ${TAG}_cfg4 data-efficient-b32
timeout 600 python tools/loop_bench.py > gpurun_out/${TAG}_loop_bench.json.log 2>&1; tail -1 gpurun_out/${TAG}_loop_bench.json.log | cut -c1-600
