# usage: bash tools/gpu_evidence.sh <tag> <config>
# evidence set of one bench config: the bench line, rocprofv3 kernel stats, two --pmc passes (FETCH_SIZE / WRITE_SIZE, kernel-trace
# only) summarised per bench tag -> gpurun_out/<tag>_*  (copy into profiles/ to commit)
TAG=$1; CFG=$2
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$PWD
timeout 900 python bench.py --config $CFG > gpurun_out/${TAG}_bench.json.log 2>&1; echo "bench rc=$?"
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_prof -o ${TAG} -- python $ROOT/bench.py --config $CFG --steps 300 --warmup 50 --no-cpu-baseline --no-profile > $ROOT/gpurun_out/${TAG}_prof.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $ROOT/gpurun_out/${TAG}_pmc_$C -o pmc -- python $ROOT/bench.py --config $CFG --steps 40 --warmup 10 --no-cpu-baseline --no-profile > $ROOT/gpurun_out/${TAG}_pmc_$C.log 2>&1
done
cd $ROOT
python tools/pmc_summary.py gpurun_out/${TAG}_pmc_FETCH_SIZE gpurun_out/${TAG}_pmc_WRITE_SIZE > gpurun_out/${TAG}_pmc.json
cp $(find gpurun_out/${TAG}_prof -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_kernel_stats.csv
find gpurun_out/${TAG}_prof gpurun_out/${TAG}_pmc_FETCH_SIZE gpurun_out/${TAG}_pmc_WRITE_SIZE -name "*kernel_trace*" -delete 2>/dev/null
find gpurun_out/${TAG}_pmc_FETCH_SIZE gpurun_out/${TAG}_pmc_WRITE_SIZE -name "*counter_collection*" -delete 2>/dev/null
tail -1 gpurun_out/${TAG}_bench.json.log | cut -c1-400; python -c "
import json; d=json.load(open('gpurun_out/${TAG}_pmc.json')); print({k: round(v['hbm_bytes_per_launch']/1e6,2) for k,v in d.items()})"
