# round 5, last GPU call: the evidence that carries the library's source hash, regenerated on the head WITHOUT the counter passes
# (a `rocprofv3 --pmc` pass hung for its whole 900 s limit three times in a row on the previous call's box after the kernel-stats pass
# of the same command line had finished normally; the PMC traffic files under profiles/ are those of the build before the split pass
# was removed — the kernels they describe are unchanged).  Every step under `timeout -k`.
TAG=${1:-round5_final}
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$PWD
timeout -k 10 400 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_gpu.log; tail -2 gpurun_out/${TAG}_pytest_gpu.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/${TAG}_smoke.log; tail -2 gpurun_out/${TAG}_smoke.log
for pair in "cfg2 pong-canonical-b32" "cfg3 breakout-canonical-b256" "cfg4 data-efficient-b32"; do
  set -- $pair
  timeout -k 10 300 python bench.py --config $2 > gpurun_out/${TAG}_$1_bench.json.log 2>&1; echo "bench $1 rc=$?"
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_$1_prof -o ${TAG}_$1 -- python $ROOT/bench.py --config $2 --steps 300 --warmup 50 --no-cpu-baseline --no-profile > $ROOT/gpurun_out/${TAG}_$1_prof.log 2>&1)
  cp $(find gpurun_out/${TAG}_$1_prof -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_$1_kernel_stats.csv 2>/dev/null
  rm -rf gpurun_out/${TAG}_$1_prof
  tail -1 gpurun_out/${TAG}_$1_bench.json.log | cut -c1-160
done
bash tools/gpu_trace_gaps.sh pong-canonical-b32 > gpurun_out/${TAG}_trace.txt 2>&1; grep "n/step" gpurun_out/${TAG}_trace.txt | cut -c1-100
bash tools/gpu_trace_gaps.sh breakout-canonical-b256 > gpurun_out/${TAG}_trace_b256.txt 2>&1
timeout -k 10 300 python tools/loop_bench.py > gpurun_out/${TAG}_loop_bench.json.log 2>&1; tail -1 gpurun_out/${TAG}_loop_bench.json.log | cut -c1-200
for v in "RAINBOW_AMD_LAZY_PRIORITIES=0" "RAINBOW_AMD_LAZY_PRIORITIES=1 PER_DONATE=0" "RAINBOW_AMD_LAZY_PRIORITIES=1 PER_DONATE=1"; do echo "[$v]"; env $v timeout -k 10 120 python tools/per_bench.py 2>/dev/null | tail -1; done > gpurun_out/${TAG}_per_bench.txt
timeout -k 10 120 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/${TAG}_20step_bench.json.log
