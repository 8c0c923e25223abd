set -x
mkdir -p gpurun_out
for C in breakout-canonical-b256 data-efficient-b32; do
  timeout 600 python bench.py --config $C --steps 300 --warmup 50 --no-cpu-baseline > gpurun_out/cfg_$C.log 2>&1
  tail -1 gpurun_out/cfg_$C.log | cut -c1-600
done
