# usage: bash tools/gpu_envsweep.sh "A=1 B=2|A=0 B=2|..." [bench args]: alternating rounds of bench.py under each '|'-separated environment
SETS=$1; shift
mkdir -p gpurun_out
IFS='|' read -ra ARR <<< "$SETS"
for R in 1 2 3; do
  i=0
  for E in "${ARR[@]}"; do
    i=$((i+1))
    env $E timeout 600 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline "$@" > gpurun_out/sweep_${i}_$R.log 2>&1
    python - "$E" $R gpurun_out/sweep_${i}_$R.log <<'PY'
import json, sys
l=[x for x in open(sys.argv[3]) if x.startswith("{")]
d=json.loads(l[-1]); r=d["roofline"]
ks={o["kernel"]: o["avg_us"] for o in d["roofline_others"]}; ks[r["kernel"]]=r["avg_us"]; ks.update(d.get("extra_kernel_us", {}))
print("[%s] r%s: %.1f steps/s %.1f us | %s" % (sys.argv[1], sys.argv[2], d["value"], d["ms_per_step_unbracketed"]*1e3, " ".join("%s %.1f" % (k, v) for k, v in sorted(ks.items()))))
PY
  done
done
