# usage: bash tools/gpu_r6_final.sh — the round's evidence set (tools/gpu_r6_evidence.sh) + the per-workgroup timelines of the head on an -DRB_STAMP
# build (bash tools/build_variant.sh stamp -DRB_STAMP first) + soak runs of the other two configs (the in-launch write-back path: ring wraps, target syncs)
bash tools/gpu_r6_evidence.sh round6_final
RAINBOW_AMD_LIB=$PWD/rainbow_amd/librainbow_hip_stamp.so timeout -k 10 300 python tools/wg_timeline.py pong-canonical-b32 > gpurun_out/round6_final_wg_timeline_b32.txt 2>&1
RAINBOW_AMD_LIB=$PWD/rainbow_amd/librainbow_hip_stamp.so timeout -k 10 300 python tools/wg_timeline.py data-efficient-b32 > gpurun_out/round6_final_wg_timeline_cfg4.txt 2>&1
SOAK_CONFIG=data-efficient-b32 SOAK_STEPS=40000 timeout -k 10 600 python tools/soak.py > gpurun_out/round6_final_soak_cfg4.json.log 2>&1; tail -1 gpurun_out/round6_final_soak_cfg4.json.log | cut -c1-200
SOAK_CONFIG=breakout-canonical-b256 SOAK_STEPS=20000 timeout -k 10 600 python tools/soak.py > gpurun_out/round6_final_soak_cfg3.json.log 2>&1; tail -1 gpurun_out/round6_final_soak_cfg3.json.log | cut -c1-200
