# usage: bash tools/gpu_check.sh <tag> [bench args]   — GPU tests + smoke + bench + rocprof kernel stats
set -x
TAG=$1; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --durations=12 > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/${TAG}_smoke.log
timeout 600 python bench.py --steps 500 --warmup 100 --no-cpu-baseline "$@" > gpurun_out/${TAG}_bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/${TAG}_bench.log
ROOT=$PWD
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_prof -o ${TAG} -- python $ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline "$@" > $ROOT/gpurun_out/${TAG}_prof.log 2>&1
cd $ROOT
find gpurun_out/${TAG}_prof -name "*kernel_trace*" -delete 2>/dev/null
ls -R gpurun_out/${TAG}_prof | head
tail -25 gpurun_out/${TAG}_pytest.log; tail -2 gpurun_out/${TAG}_smoke.log; tail -2 gpurun_out/${TAG}_bench.log | cut -c1-900
