# rocprof kernel stats of the one-rank replica-exchange plumbing run (RAINBOW_AMD_FORCE_DIST=1); $1 = factored|allreduce
MODE=${1:-factored}
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp && RAINBOW_AMD_EXCHANGE=$MODE RAINBOW_AMD_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29577 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/dist1_${MODE}_prof -o d1 -- python $ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $ROOT/gpurun_out/dist1_${MODE}_prof.log 2>&1
cd $ROOT
find gpurun_out/dist1_${MODE}_prof -name "*kernel_trace*" -delete 2>/dev/null
tail -1 gpurun_out/dist1_${MODE}_prof.log | cut -c1-200
