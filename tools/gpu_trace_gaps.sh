# usage: bash tools/gpu_trace_gaps.sh <config>: rocprofv3 kernel trace of a short bench run; per kernel of the learn step the
# mean duration and the mean idle gap between the previous kernel's end and its start (steady-state steps only).
# bench.py runs with --no-profile: a bracketed launch (HIP event records = system-scope barrier packets) shows ~6 us of idle
# GPU on both sides, and the LAST steps of a default bench run are its bracketed roofline_others passes (round 2 read those
# as 'two idle gaps at the sampler launch': the sampler was simply the last tag bracketed)
CFG=$1
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/gaps -o gaps -- python $ROOT/bench.py --config $CFG --steps 60 --warmup 20 --no-cpu-baseline --no-profile > $ROOT/gpurun_out/gaps.log 2>&1
cd $ROOT
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/gaps/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the timed region: the last 40 k_head launches delimit steps (one per learn call; the optimiser pass may be hosted by the
# sampler launch, RB_LEARNER_DEFER_UPDATE, so k_clip_adam is not a per-step launch any more)
idx = [i for i, r in enumerate(rows) if "k_head<" in r["Kernel_Name"] or r["Kernel_Name"].startswith("k_head(")]
lo, hi = idx[-41], idx[-1]
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
for i in range(lo + 1, hi + 1):
    r, p = rows[i], rows[i - 1]
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:56] + ("|" + str(r.get("Grid_Size") or r.get("Grid_Size_X")) if "k_nl_" in r["Kernel_Name"] else "")
    dur[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    gap[name].append((int(r["Start_Timestamp"]) - int(p["End_Timestamp"])) / 1e3)
# the PER-only phase of bench.py (k_sample / k_update alternate, no learner): the same statistics
pidx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_update")]
if len(pidx) > 100:
    d2 = collections.defaultdict(list); g2 = collections.defaultdict(list)
    for i in range(pidx[50], pidx[-50]):
        r, p = rows[i], rows[i - 1]
        name = "PER phase: " + r["Kernel_Name"].split("(")[0][:40]
        d2[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        g2[name].append((int(r["Start_Timestamp"]) - int(p["End_Timestamp"])) / 1e3)
    for k in d2:
        print("%-60s n %5d  dur %7.2f us  gap-before %6.2f us" % (k, len(d2[k]), sum(d2[k]) / len(d2[k]), sum(g2[k]) / len(g2[k])))
tot = 0
for k in dur:
    d, g = sum(dur[k]) / len(dur[k]), sum(gap[k]) / len(gap[k])
    print("%-60s n/step %.1f  dur %7.2f us  gap-before %6.2f us" % (k, len(dur[k]) / 40.0, d, g))
print("columns:", list(rows[0].keys()))
for i in range(hi - 15, hi + 1):
    r = rows[i]
    print({k: r[k] for k in r if k in ("Queue_Id", "Stream_Id", "Thread_Id", "Dispatch_Id", "Agent_Id", "Correlation_Id", "Private_Segment_Size", "LDS_Block_Size", "Scratch_Size")}, r["Kernel_Name"][:30])
PY
rm -rf gpurun_out/gaps
