# rocprof kernel stats of the acting loop only (tools/loop_bench.py with LOOP_ONLY_ACT=1)
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp && LOOP_ONLY_ACT=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/act_prof -o act -- python $ROOT/tools/loop_bench.py > $ROOT/gpurun_out/act_prof.log 2>&1
cd $ROOT
find gpurun_out/act_prof -name "*kernel_trace*" -delete 2>/dev/null
tail -2 gpurun_out/act_prof.log
