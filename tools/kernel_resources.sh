# usage: bash tools/kernel_resources.sh [pattern] — VGPRs / spills / LDS / occupancy of every kernel of the library whose name matches
PAT=${1:-.}
cd rainbow_amd/csrc
for f in learner.hip replay.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --cuda-device-only -c $f -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
  python3 -c "
import re, sys
cur = None; rows = {}
for line in sys.stdin:
    m = re.search(r'remark: (?:Function Name: (\S+)|\s+([A-Za-z ]+(?:\[[^\]]*\])?): (\S+))', line)
    if not m: continue
    if m.group(1): cur = m.group(1); rows[cur] = {}
    elif cur: rows[cur][m.group(2).strip()] = m.group(3)
import subprocess
for k, v in rows.items():
    name = subprocess.run(['c++filt', k], capture_output=True, text=True).stdout.strip()
    if not re.search(sys.argv[1], name): continue
    print('%-110s VGPR %4s AGPR %3s spill v%s s%s scratch %s LDS %6s occ %s' % (name[:110], v.get('VGPRs'), v.get('AGPRs'), v.get('VGPRs Spill'), v.get('SGPRs Spill'), v.get('ScratchSize [bytes/lane]'), v.get('LDS Size [bytes/block]'), v.get('Occupancy [waves/SIMD]')))
" "$PAT"
done
