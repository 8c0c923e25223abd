set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo | grep -E "Marketing|gfx" | head -4 > gpurun_out/r1_env.txt 2>&1
nproc >> gpurun_out/r1_env.txt; lscpu | grep "Model name" >> gpurun_out/r1_env.txt
timeout 120 python -c "import __graft_entry__ as g; g.build(); print('build ok')" > gpurun_out/r1_build.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r1_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r1_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r1_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r1_smoke.log
timeout 600 python bench.py --steps 300 --warmup 50 --no-cpu-baseline > gpurun_out/r1_bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/r1_bench.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r1_prof -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 20 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r1_prof.log 2>&1
cd $GRAFT_REPO_ROOT; ls -R gpurun_out/r1_prof | head -20
tail -5 gpurun_out/r1_pytest.log; tail -3 gpurun_out/r1_smoke.log; tail -3 gpurun_out/r1_bench.log
