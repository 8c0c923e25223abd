# usage: bash tools/gpu_r6_evidence.sh <tag>: round 6's evidence set (copy gpurun_out/<tag>_* into profiles/ to commit): GPU suite (plain and with
# serialised blocking launches), smoke, per BASELINE config the bench line + rocprofv3 kernel stats + FETCH / WRITE traffic + SQ counters, the
# kernel traces of configs 2, 3 and 4, the driver's 20-step line, the loop bench, the PER bench, the exchange kernels of config 5 on one device,
# the soak run in the shape of main.py's loop.
# Every step under `timeout -k`.
TAG=${1:-round6_final}
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$PWD
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_gpu.log
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" gpurun_out/${TAG}_pytest_gpu.log | tail -3
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/${TAG}_smoke.log; tail -2 gpurun_out/${TAG}_smoke.log
timeout -k 10 120 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/${TAG}_20step_bench.json.log; cut -c1-160 gpurun_out/${TAG}_20step_bench.json.log
for pair in "cfg2 pong-canonical-b32" "cfg3 breakout-canonical-b256" "cfg4 data-efficient-b32"; do
  set -- $pair
  timeout -k 10 400 python bench.py --config $2 > gpurun_out/${TAG}_$1_bench.json.log 2>&1; echo "bench $1 rc=$?"; tail -1 gpurun_out/${TAG}_$1_bench.json.log | cut -c1-160
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_$1_prof -o ${TAG}_$1 -- python $ROOT/bench.py --config $2 --steps 300 --warmup 50 --no-cpu-baseline --no-profile > $ROOT/gpurun_out/${TAG}_$1_prof.log 2>&1)
  cp $(find gpurun_out/${TAG}_$1_prof -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_$1_kernel_stats.csv 2>/dev/null
  rm -rf gpurun_out/${TAG}_$1_prof
  for C in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout -k 10 240 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $ROOT/gpurun_out/${TAG}_$1_pmc_$C -o pmc -- python $ROOT/bench.py --config $2 --steps 40 --warmup 10 --no-cpu-baseline --no-profile > $ROOT/gpurun_out/${TAG}_$1_pmc_$C.log 2>&1); echo "pmc $C $1 rc=$?"
  done
  python tools/pmc_summary.py gpurun_out/${TAG}_$1_pmc_FETCH_SIZE gpurun_out/${TAG}_$1_pmc_WRITE_SIZE > gpurun_out/round6_pmc_$2.json 2>/dev/null
  rm -rf gpurun_out/${TAG}_$1_pmc_FETCH_SIZE gpurun_out/${TAG}_$1_pmc_WRITE_SIZE
  bash tools/gpu_sqpmc.sh $2 round6 2>&1 | grep mfma_busy | cut -c1-150
done
bash tools/gpu_trace_gaps.sh pong-canonical-b32 > gpurun_out/${TAG}_trace.txt 2>&1; grep "n/step" gpurun_out/${TAG}_trace.txt | cut -c1-100
bash tools/gpu_trace_gaps.sh breakout-canonical-b256 > gpurun_out/${TAG}_trace_b256.txt 2>&1
bash tools/gpu_trace_gaps.sh data-efficient-b32 > gpurun_out/${TAG}_trace_cfg4.txt 2>&1
timeout -k 10 300 python tools/loop_bench.py > gpurun_out/${TAG}_loop_bench.json.log 2>&1; tail -1 gpurun_out/${TAG}_loop_bench.json.log | cut -c1-200
for v in "RAINBOW_AMD_LAZY_PRIORITIES=0" "RAINBOW_AMD_LAZY_PRIORITIES=1 PER_DONATE=0" "RAINBOW_AMD_LAZY_PRIORITIES=1 PER_DONATE=1"; do echo "[$v]"; env $v timeout -k 10 120 python tools/per_bench.py 2>/dev/null | tail -1; done > gpurun_out/${TAG}_per_bench.txt
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_exch -o exch -- python -m pytest $ROOT/tests/test_exchange_gpu.py -q -k "factored and 8" -p no:cacheprovider > $ROOT/gpurun_out/${TAG}_exch.log 2>&1)
python tools/exchange_world8_times.py $(find gpurun_out/${TAG}_exch -name "*kernel_stats.csv" | head -1) > gpurun_out/${TAG}_exchange_world8.txt 2>&1; cat gpurun_out/${TAG}_exchange_world8.txt
rm -rf gpurun_out/${TAG}_exch gpurun_out/${TAG}_exch.log
SOAK_STEPS=40000 timeout -k 10 600 python tools/soak.py > gpurun_out/${TAG}_soak.json.log 2>&1; tail -1 gpurun_out/${TAG}_soak.json.log | cut -c1-300
AMD_SERIALIZE_KERNEL=3 HIP_LAUNCH_BLOCKING=1 timeout -k 10 1200 python -m pytest tests -m gpu -q -x > gpurun_out/${TAG}_serialized_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_serialized_pytest_gpu.log
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" gpurun_out/${TAG}_serialized_pytest_gpu.log | tail -3
