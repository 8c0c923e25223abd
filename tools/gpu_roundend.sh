# usage: bash tools/gpu_roundend.sh <tag>: the evidence set committed under profiles/ at the end of a round —
# GPU test log, smoke, and per bench config: bench line (with cpu_baseline), rocprofv3 kernel stats, PMC traffic; the per-kernel
# trace of the headline config (durations + idle gaps), the loop bench, the per-workgroup timeline (RB_STAMP build when present)
TAG=$1
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_gpu.log; grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" gpurun_out/${TAG}_pytest_gpu.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/${TAG}_smoke.log; tail -2 gpurun_out/${TAG}_smoke.log
bash tools/gpu_evidence.sh ${TAG}_cfg2 pong-canonical-b32
bash tools/gpu_trace_gaps.sh pong-canonical-b32 > gpurun_out/${TAG}_trace.txt 2>&1; grep "n/step" gpurun_out/${TAG}_trace.txt | cut -c1-120
bash tools/gpu_evidence.sh ${TAG}_cfg3 breakout-canonical-b256
bash tools/gpu_evidence.sh ${TAG}_cfg4 data-efficient-b32
timeout 600 python tools/loop_bench.py > gpurun_out/${TAG}_loop_bench.json.log 2>&1; tail -1 gpurun_out/${TAG}_loop_bench.json.log | cut -c1-600
if [ -f rainbow_amd/librainbow_hip_stamp.so ]; then
  RAINBOW_AMD_LIB=$PWD/rainbow_amd/librainbow_hip_stamp.so timeout 200 python tools/wg_timeline.py > gpurun_out/${TAG}_wg_timeline.txt 2>&1
  RAINBOW_AMD_LIB=$PWD/rainbow_amd/librainbow_hip_stamp.so timeout 100 python tools/stamp/act_timeline.py > gpurun_out/${TAG}_act_timeline.txt 2>&1
fi
if [ -f rainbow_amd/librainbow_hip_stamp.so ]; then
  RAINBOW_AMD_LIB=$PWD/rainbow_amd/librainbow_hip_stamp.so timeout 120 python tools/stamp/gemm_timeline.py 2>&1 | grep "==\|   " > gpurun_out/${TAG}_gemm_timeline.txt
fi
for v in "RAINBOW_AMD_LAZY_PRIORITIES=0" "RAINBOW_AMD_LAZY_PRIORITIES=1 PER_DONATE=0" "RAINBOW_AMD_LAZY_PRIORITIES=1 PER_DONATE=1"; do echo "[$v]"; env $v timeout 120 python tools/per_bench.py 2>/dev/null | tail -1; done > gpurun_out/${TAG}_per_bench.txt
for v in "RB_OPTS=fc_gemm=0" "RB_OPTS=fc_gemm=-1" "RB_OPTS=fc_gemm=0" "RB_OPTS=fc_gemm=-1"; do
  env $v timeout 90 python bench.py --config breakout-canonical-b256 --steps 1000 --warmup 200 --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read() or '{\"ms_per_step\":0,\"value\":0}'); print('[$v]: %.2f us/step  %.0f steps/s' % (d['ms_per_step']*1e3, d['value']))"
done > gpurun_out/${TAG}_fc_gemm_ab.txt
bash tools/gpu_trace_gaps.sh breakout-canonical-b256 > gpurun_out/${TAG}_trace_b256.txt 2>&1
timeout 120 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/${TAG}_20step_bench.json.log
