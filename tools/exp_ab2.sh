export TMPDIR=/tmp
ROOT=$PWD
echo "== bias prefetch (variant nobiaspre = old epilogue loads) vs default" | tee gpurun_out/round6_biaspre_ab.txt
for cfg in pong-canonical-b32 data-efficient-b32; do
CFG=$cfg ROUNDS=3 bash tools/gpu_env_ab.sh "RAINBOW_AMD_LIB=$ROOT/rainbow_amd/librainbow_hip_nobiaspre.so" "RAINBOW_AMD_LIB=$ROOT/rainbow_amd/librainbow_hip.so" 2>&1 | sed "s/^/$cfg /; s#$ROOT/rainbow_amd/##" | tee -a gpurun_out/round6_biaspre_ab.txt
done
echo "== z_ct / h_ct" | tee gpurun_out/round6_ct_ab.txt
for cfg in pong-canonical-b32 data-efficient-b32; do
CFG=$cfg ROUNDS=2 bash tools/gpu_env_ab.sh "RB_OPTS=z_ct=2" "RB_OPTS=z_ct=1" "RB_OPTS=h_ct=2" "RB_OPTS=h_ct=8" "RB_OPTS=z_ct=1,h_ct=2" 2>&1 | sed "s/^/$cfg /" | tee -a gpurun_out/round6_ct_ab.txt
done
PYTEST_K="baseline_shapes or golden" RB_OPTS="z_ct=1,h_ct=2" timeout 600 python -m pytest tests -m gpu -q -k "baseline_shapes or golden" 2>&1 | tail -3
