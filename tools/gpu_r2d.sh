mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/r2d_pytest.log 2>&1; tail -3 gpurun_out/r2d_pytest.log
for M in lds global; do echo "== RB_SAMPLER=$M"; RB_SAMPLER=$M python tools/sample_bench.py 2>&1 | tail -1; RB_SAMPLER=$M SAMPLE_CONFIG=data-efficient-b32 python tools/sample_bench.py 2>&1 | tail -1; RB_SAMPLER=$M SAMPLE_CONFIG=breakout-canonical-b256 python tools/sample_bench.py 2>&1 | tail -1; done
bash tools/gpu_ab.sh RB_SAMPLER "lds global" --extra-tags fc_z_bwd,fc_z_fwd 2>&1 | grep -E "round|sample|fc_h_bwd|fc_z"
