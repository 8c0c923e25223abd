# HBM traffic of the dominant kernel from PMC counters, one counter per pass (MI355X_MICROARCH.md §HBM:
# FETCH_SIZE and WRITE_SIZE do not fit one pass; --pmc runs carry --kernel-trace only).
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_$C -o pmc -- python $ROOT/bench.py --steps 40 --warmup 10 --no-cpu-baseline > $ROOT/gpurun_out/pmc_$C.log 2>&1
done
cd $ROOT
python tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE > gpurun_out/pmc_summary.json
cat gpurun_out/pmc_summary.json
find gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE -name "*kernel_trace*" -delete
ls -la gpurun_out/pmc_FETCH_SIZE | head
