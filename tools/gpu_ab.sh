# usage: bash tools/gpu_ab.sh VAR "v1 v2 ..." [bench args]  — same-box A/B of an environment switch (3 alternating rounds)
VAR=$1; VALS=$2; shift; shift
mkdir -p gpurun_out
for R in 1 2 3; do
  for V in $VALS; do
    env $VAR=$V timeout 600 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline "$@" > gpurun_out/ab_${VAR}_${V}_$R.log 2>&1
    python - <<PY
import json
l=[x for x in open("gpurun_out/ab_${VAR}_${V}_$R.log") if x.startswith("{")]
d=json.loads(l[-1]); r=d["roofline"]
print("$VAR=$V round $R: %.1f steps/s  %.1f us/step (unbracketed %.1f)  %s %.1f us frac %.3f" % (d["value"], d["ms_per_step"]*1e3, d["ms_per_step_unbracketed"]*1e3, r["kernel"], r["avg_us"], r["frac"]))
for o in d["roofline_others"]: print("      %-12s %6.1f us  %.3f" % (o["kernel"], o["avg_us"], o["frac"]))
for k, v in d.get("extra_kernel_us", {}).items(): print("      %-12s %6.1f us" % (k, v))
PY
  done
done
