# usage: bash tools/gpu_sqpmc.sh [config] [tag] — one rocprofv3 --pmc pass with eight SQ counters (kernel-trace only, as gpurun requires) over a
# short bench run of `config`, summarised per kernel (tools/sq_summary.py) -> gpurun_out/<tag>_sq_counters_<config>.json
CFG=${1:-pong-canonical-b32}
TAG=${2:-sq}
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_sq_$CFG -o pmc -- python $ROOT/bench.py --config $CFG --steps 30 --warmup 10 --no-cpu-baseline --no-profile > $ROOT/gpurun_out/pmc_sq_$CFG.log 2>&1
cd $ROOT
python tools/sq_summary.py gpurun_out/pmc_sq_$CFG > gpurun_out/${TAG}_sq_counters_$CFG.json
rm -rf gpurun_out/pmc_sq_$CFG
tail -2 gpurun_out/pmc_sq_$CFG.log | cut -c1-200
python - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_sq_counters_$CFG.json"))
for k, v in d.items():
    if v["launches"] >= 20 and v["mfma_busy_cycles"] > 0:
        print("%-70s mfma_busy %.3f  wait %.2f  inst-stall %.2f  issue %.2f  lds-conflict %.2f" % (k[:70], v["mfma_busy_frac"], v["wait_any_frac"], v["wait_inst_frac"], v["active_inst_frac"], v["lds_conflict_frac_of_lds_active"]))
PY
