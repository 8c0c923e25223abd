set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
for A in 0; do
  RB_ABLATE=$A timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/abl_$A -o a -- python $ROOT/bench.py --steps 150 --warmup 20 --no-cpu-baseline > $ROOT/gpurun_out/abl_$A.log 2>&1
  find $ROOT/gpurun_out/abl_$A -name "*kernel_trace*" -delete
done
cd $ROOT
python - <<'PY'
import csv,glob
for A in [0]:
    f=glob.glob('gpurun_out/abl_%d/**/*kernel_stats.csv'%A, recursive=True)[0]
    out=[]
    for r in csv.DictReader(open(f)):
        if True:
            out.append('%s=%.1f' % (r['Name'].split('<')[1][:22] if '<' in r['Name'] else 'head', float(r['AverageNs'])/1e3))
    print('ablate',A,' '.join(sorted(out)))
PY
