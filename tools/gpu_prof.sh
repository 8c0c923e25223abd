# usage: bash tools/gpu_prof.sh <tag> [bench args] — rocprofv3 kernel stats of bench.py (per-kernel mean durations)
TAG=$1; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_prof -o ${TAG} -- python $ROOT/bench.py --steps 300 --warmup 50 --no-cpu-baseline --roofline-kernel clip_adam "$@" > $ROOT/gpurun_out/${TAG}_prof.log 2>&1
cd $ROOT
find gpurun_out/${TAG}_prof -name "*kernel_trace*" -delete 2>/dev/null
python - $TAG <<'PY'
import csv, glob, sys
f = glob.glob('gpurun_out/%s_prof/**/*kernel_stats.csv' % sys.argv[1], recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 300 <= int(r['Calls']) <= 4000 and not r['Name'].startswith(('void at::', 'void (anonymous'))]
tot = 0.0
for r in rows:
    per_step = float(r['AverageNs']) * int(r['Calls']) / 1e3
    print("%-70s calls %5s avg %7.2f us" % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3))
PY
