export TMPDIR=/tmp; ROOT=$PWD; mkdir -p gpurun_out; cd /tmp
for v in "RAINBOW_AMD_IMPLICIT_SIGMA=1" "RAINBOW_AMD_IMPLICIT_SIGMA=0"; do
  env $v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/adam_sa -o sa -- python $ROOT/tools/exp/adam_standalone.py > $ROOT/gpurun_out/adam_sa.log 2>&1
  echo "[$v]"; python - <<PY
import csv, glob
f = glob.glob("$ROOT/gpurun_out/adam_sa/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r["Name"] for k in ("k_adam_pending", "k_clip_adam", "k_sample<")):
        print("  %-40s calls %5s avg %7.2f us" % (r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  rm -rf $ROOT/gpurun_out/adam_sa
done
