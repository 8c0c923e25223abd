set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
for V in 256 512 1024 2048; do
  RB_HS=$V timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/sw_$V -o a -- python $ROOT/bench.py --steps 150 --warmup 20 --no-cpu-baseline > $ROOT/gpurun_out/sw_$V.log 2>&1
  find $ROOT/gpurun_out/sw_$V -name "*kernel_trace*" -delete
done
cd $ROOT
python - <<'PY'
import csv,glob,json
for V in [256,512,1024,2048]:
    f=glob.glob('gpurun_out/sw_%d/**/*kernel_stats.csv'%V, recursive=True)[0]
    out=[]
    for r in csv.DictReader(open(f)):
        if 'k_nl_fwd' in r['Name'] or 'k_fc_h_finish' in r['Name'] or 'conv_dx' in r['Name']:
            out.append('%s avg=%.1f max=%.1f' % (r['Name'][:28], float(r['AverageNs'])/1e3, float(r['MaxNs'])/1e3))
    d=json.loads(open('gpurun_out/sw_%d.log'%V).read().strip().split('\n')[-1])
    print('target_blocks',V, round(d['value']), 'steps/s |', ' | '.join(sorted(out)))
PY
