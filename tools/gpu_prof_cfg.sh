# usage: bash tools/gpu_prof_cfg.sh <config> <tag>  — rocprof kernel stats of one bench config
CFG=$1; TAG=$2
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_prof -o ${TAG} -- python $ROOT/bench.py --config $CFG --steps 100 --warmup 20 --no-cpu-baseline > $ROOT/gpurun_out/${TAG}_prof.log 2>&1
cd $ROOT
find gpurun_out/${TAG}_prof -name "*kernel_trace*" -delete 2>/dev/null
tail -1 gpurun_out/${TAG}_prof.log | cut -c1-300
