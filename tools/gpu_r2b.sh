set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r2b_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2b_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2b_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r2b_smoke.log
timeout 600 python bench.py --steps 500 --warmup 100 --no-cpu-baseline > gpurun_out/r2b_bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/r2b_bench.log
timeout 100 python bench.py --gpus 2 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r2b_gpus2.log 2>&1; echo "gpus2 rc=$?" >> gpurun_out/r2b_gpus2.log
for C in breakout-canonical-b256 data-efficient-b32; do
  timeout 600 python bench.py --config $C --steps 300 --warmup 50 --no-cpu-baseline > gpurun_out/r2b_$C.log 2>&1
done
RAINBOW_AMD_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29555 bench.py --steps 300 --warmup 50 --no-cpu-baseline > gpurun_out/r2b_dist1.log 2>&1
tail -12 gpurun_out/r2b_pytest.log; tail -2 gpurun_out/r2b_smoke.log; for f in r2b_bench r2b_gpus2 r2b_breakout-canonical-b256 r2b_data-efficient-b32 r2b_dist1; do tail -2 gpurun_out/$f.log | cut -c1-300; done
