bash tools/gpu_r6_evidence.sh round6_final
RAINBOW_AMD_LIB=$PWD/rainbow_amd/librainbow_hip_stamp.so timeout -k 10 300 python tools/wg_timeline.py pong-canonical-b32 > gpurun_out/round6_final_wg_timeline_b32.txt 2>&1
RAINBOW_AMD_LIB=$PWD/rainbow_amd/librainbow_hip_stamp.so timeout -k 10 300 python tools/wg_timeline.py data-efficient-b32 > gpurun_out/round6_final_wg_timeline_cfg4.txt 2>&1
