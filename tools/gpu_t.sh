mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_learner_gpu.py -m gpu -x -q -k graph 2>&1 | tail -60 > gpurun_out/t.log
