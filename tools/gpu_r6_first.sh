# round 6, first GPU call: the GPU suite on the new head (early draw opt-in + fail-safe waits), the headline bench line, the same-box
# A/B of the early draw on the three BASELINE configs, and the counter passes that hung in round 5 (now with the early draw off by
# default: FETCH_SIZE / WRITE_SIZE / SQ on the head, every step under `timeout -k`).
TAG=${1:-round6_first}
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$PWD
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_gpu.log
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" gpurun_out/${TAG}_pytest_gpu.log | tail -4
timeout -k 10 300 python bench.py > gpurun_out/${TAG}_cfg2_bench.json.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/${TAG}_cfg2_bench.json.log | cut -c1-300
timeout -k 10 120 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/${TAG}_20step_bench.json.log; cut -c1-200 gpurun_out/${TAG}_20step_bench.json.log
for cfg in pong-canonical-b32 breakout-canonical-b256 data-efficient-b32; do
  CFG=$cfg ROUNDS=2 bash tools/gpu_env_ab.sh "RB_OPTS=spec_draw=0" "RB_OPTS=spec_draw=1" 2>&1 | sed "s/^/$cfg /"
done | tee gpurun_out/${TAG}_spec_draw_ab.txt
# counters on the head (the passes that hung in round 5 with the early draw on)
for pair in "cfg2 pong-canonical-b32"; do
  set -- $pair
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_$1_prof -o ${TAG}_$1 -- python $ROOT/bench.py --config $2 --steps 300 --warmup 50 --no-cpu-baseline --no-profile > $ROOT/gpurun_out/${TAG}_$1_prof.log 2>&1)
  cp $(find gpurun_out/${TAG}_$1_prof -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_$1_kernel_stats.csv 2>/dev/null
  rm -rf gpurun_out/${TAG}_$1_prof
  for C in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout -k 10 240 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $ROOT/gpurun_out/${TAG}_$1_pmc_$C -o pmc -- python $ROOT/bench.py --config $2 --steps 40 --warmup 10 --no-cpu-baseline --no-profile > $ROOT/gpurun_out/${TAG}_$1_pmc_$C.log 2>&1); echo "pmc $C $1 rc=$?"
  done
  python tools/pmc_summary.py gpurun_out/${TAG}_$1_pmc_FETCH_SIZE gpurun_out/${TAG}_$1_pmc_WRITE_SIZE > gpurun_out/${TAG}_$1_pmc.json 2>/dev/null
  rm -rf gpurun_out/${TAG}_$1_pmc_FETCH_SIZE gpurun_out/${TAG}_$1_pmc_WRITE_SIZE
  python -c "
import json; d=json.load(open('gpurun_out/${TAG}_$1_pmc.json')); print({k: round(v['hbm_bytes_per_launch']/1e6,2) for k,v in d.items()})" 2>/dev/null
  bash tools/gpu_sqpmc.sh $2 ${TAG} 2>&1 | tail -16
done
# the same counter pass WITH the early draw on (what hung): bounded waits must let it terminate
(cd /tmp && RB_OPTS=spec_draw=1 timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $ROOT/gpurun_out/${TAG}_spec_pmc -o pmc -- python $ROOT/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-profile > $ROOT/gpurun_out/${TAG}_spec_pmc.log 2>&1); echo "pmc with spec_draw=1 rc=$?"
tail -1 gpurun_out/${TAG}_spec_pmc.log | cut -c1-200
rm -rf gpurun_out/${TAG}_spec_pmc
bash tools/gpu_trace_gaps.sh pong-canonical-b32 > gpurun_out/${TAG}_trace.txt 2>&1; grep "n/step" gpurun_out/${TAG}_trace.txt | cut -c1-110
