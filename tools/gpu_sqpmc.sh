# one rocprofv3 --pmc pass with eight SQ counters (kernel-trace only, as gpurun requires), summarised per kernel
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_sq -o pmc -- python $ROOT/bench.py --steps 30 --warmup 10 --no-cpu-baseline > $ROOT/gpurun_out/pmc_sq.log 2>&1
cd $ROOT
python tools/sq_summary.py gpurun_out/pmc_sq > gpurun_out/sq_summary.json
find gpurun_out/pmc_sq -name "*kernel_trace*" -delete; find gpurun_out/pmc_sq -name "*counter_collection*" -delete
tail -3 gpurun_out/pmc_sq.log | cut -c1-200
head -c 3000 gpurun_out/sq_summary.json
