# round 6, third GPU call: GPU suite; the folded reduction with the READY flag (was +20 us: a thousand pollers on the arrival counter);
# the opt-in early draw at batch 256 with the LDS-free gate (was 537 vs 498 us); the exchange kernels of config 5 timed on one device
TAG=${1:-round6_third}
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$PWD
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_gpu.log
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" gpurun_out/${TAG}_pytest_gpu.log | tail -6
for cfg in pong-canonical-b32 breakout-canonical-b256 data-efficient-b32; do
  CFG=$cfg ROUNDS=3 bash tools/gpu_env_ab.sh "RB_OPTS=fold_reduce=0,iso=0" "RB_OPTS=fold_reduce=1,iso=0" 2>&1 | sed "s/^/$cfg /"
done | tee gpurun_out/${TAG}_fold_reduce_ab.txt
for cfg in pong-canonical-b32 breakout-canonical-b256 data-efficient-b32; do
  CFG=$cfg ROUNDS=3 bash tools/gpu_env_ab.sh "RB_OPTS=fold_reduce=0,iso=0" "RB_OPTS=fold_reduce=0,iso=1" 2>&1 | sed "s/^/$cfg /"
done | tee gpurun_out/${TAG}_iso_ab.txt
RB_OPTS=fold_reduce=0,iso=1 bash tools/gpu_trace_gaps.sh breakout-canonical-b256 > gpurun_out/${TAG}_trace_b256_iso.txt 2>&1; grep "n/step" gpurun_out/${TAG}_trace_b256_iso.txt | cut -c1-110
RB_OPTS=fold_reduce=1 bash tools/gpu_trace_gaps.sh pong-canonical-b32 > gpurun_out/${TAG}_trace_fold.txt 2>&1; grep "n/step" gpurun_out/${TAG}_trace_fold.txt | cut -c1-110
for cfg in breakout-canonical-b256 data-efficient-b32; do
  CFG=$cfg ROUNDS=2 bash tools/gpu_env_ab.sh "RB_OPTS=fold_reduce=0,iso=0,spec_draw=0" "RB_OPTS=fold_reduce=0,iso=0,spec_draw=1" 2>&1 | sed "s/^/$cfg /"
done | tee gpurun_out/${TAG}_spec_draw_ab.txt
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_exch -o exch -- python -m pytest $ROOT/tests/test_exchange_gpu.py -q -k "factored and 8" -p no:cacheprovider > $ROOT/gpurun_out/${TAG}_exch.log 2>&1)
python tools/exchange_world8_times.py $(find gpurun_out/${TAG}_exch -name "*kernel_stats.csv" | head -1) > gpurun_out/${TAG}_exchange_world8.txt 2>&1; cat gpurun_out/${TAG}_exchange_world8.txt
rm -rf gpurun_out/${TAG}_exch
