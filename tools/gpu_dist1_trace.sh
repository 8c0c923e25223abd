# usage: bash tools/gpu_dist1_trace.sh [factored|allreduce] — kernel trace of the ONE-rank replica-exchange plumbing run (RAINBOW_AMD_FORCE_DIST=1):
# per kernel of a step its mean duration and the idle gap before it (is the step GPU-bound or host-bound?)
MODE=${1:-factored}
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$PWD
rm -rf $ROOT/gpurun_out/d1trace
cd /tmp && RAINBOW_AMD_EXCHANGE=$MODE RAINBOW_AMD_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29577 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/d1trace -o d1 -- python $ROOT/bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-profile > $ROOT/gpurun_out/d1trace.log 2>&1
cd $ROOT
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/d1trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_head<" in r["Kernel_Name"]]
lo, hi = idx[-41], idx[-1]
dur = collections.defaultdict(list); gap = collections.defaultdict(list); order = []
for i in range(lo + 1, hi + 1):
    r, p = rows[i], rows[i - 1]
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:52] + "|" + str(r.get("Grid_Size_X"))
    if name not in dur: order.append(name)
    dur[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    gap[name].append((int(r["Start_Timestamp"]) - int(p["End_Timestamp"])) / 1e3)
step = (int(rows[hi]["End_Timestamp"]) - int(rows[lo]["End_Timestamp"])) / 40e3
busy = sum(sum(v) for v in dur.values()) / 40
print("step %.1f us, kernels busy %.1f us, idle %.1f us" % (step, busy, step - busy))
for k in order:
    print("  %-64s n/step %.1f dur %7.2f gap-before %6.2f" % (k, len(dur[k]) / 40.0, sum(dur[k]) / len(dur[k]), sum(gap[k]) / len(gap[k])))
PY
rm -rf gpurun_out/d1trace
