# end-of-round evidence: GPU tests, smoke, bench (with cpu_baseline), rocprof kernel stats, PMC traffic
set -x
TAG=$1
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$PWD
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/${TAG}_smoke.log
timeout 900 python bench.py > gpurun_out/${TAG}_bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/${TAG}_bench.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_prof -o ${TAG} -- python $ROOT/bench.py --steps 300 --warmup 50 --no-cpu-baseline > $ROOT/gpurun_out/${TAG}_prof.log 2>&1
find $ROOT/gpurun_out/${TAG}_prof -name "*kernel_trace*" -delete
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_$C -o pmc -- python $ROOT/bench.py --steps 40 --warmup 10 --no-cpu-baseline > $ROOT/gpurun_out/pmc_$C.log 2>&1
done
cd $ROOT
python tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE > gpurun_out/pmc_summary.json
find gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE -name "*kernel_trace*" -delete
find gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE -name "*counter_collection*" -delete
tail -3 gpurun_out/${TAG}_pytest.log; tail -2 gpurun_out/${TAG}_smoke.log; tail -2 gpurun_out/${TAG}_bench.log | cut -c1-1500; cat gpurun_out/pmc_summary.json
