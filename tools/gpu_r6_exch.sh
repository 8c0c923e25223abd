# usage: bash tools/gpu_r6_exch.sh <tag>: the exchange GPU tests, then the exchange kernels' durations at world 8 with the tiled and the 16-row-tile
# finishing kernel (rocprofv3 kernel stats of tests/test_exchange_gpu.py, factored mode, world 8)
TAG=${1:-round6_exch}
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$PWD
timeout -k 10 600 python -m pytest tests/test_exchange_gpu.py tests/test_dist_agent_gpu.py -m gpu -q > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_gpu.log
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" gpurun_out/${TAG}_pytest_gpu.log | tail -3
for v in 1 0; do
  (cd /tmp && RB_OPTS=finish_tiled=$v timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_exch$v -o exch -- python -m pytest $ROOT/tests/test_exchange_gpu.py -q -k "factored and 8" -p no:cacheprovider > $ROOT/gpurun_out/${TAG}_exch$v.log 2>&1)
  echo "finish_tiled=$v" >> gpurun_out/${TAG}_exchange_world8.txt
  python tools/exchange_world8_times.py $(find gpurun_out/${TAG}_exch$v -name "*kernel_stats.csv" | head -1) >> gpurun_out/${TAG}_exchange_world8.txt 2>&1
  rm -rf gpurun_out/${TAG}_exch$v gpurun_out/${TAG}_exch$v.log
done
cat gpurun_out/${TAG}_exchange_world8.txt
