"""usage: python tools/exchange_world8_times.py <kernel_stats.csv> — per-kernel durations of the replica exchange measured on
ONE device: rocprofv3 --kernel-trace --stats over tests/test_exchange_gpu.py's world-8 case (8 learner handles standing for 8
replicas, cfg-2 shape: the all-gather is a device-side copy there).  What one GPU can measure of config 5's step: k_pack_factors
(this rank's 1.02 MB block), k_finish_grads (the 8-block rank-order fold, M = 256 gathered rows) and the clip + Adam pass behind it."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
want = ("k_pack_factors", "k_finish_grads", "k_clip_adam", "k_adam_pending", "k_sumsq", "k_reduce_conv_dw_all", "k_nl_bwd", "k_conv_dw_all")
print("%-64s %8s %12s %12s %12s" % ("kernel", "calls", "avg us", "min us", "max us"))
for r in rows:
    name = r.get("Name") or r.get("KernelName") or ""
    if any(w in name for w in want):
        print("%-64s %8s %12.2f %12.2f %12.2f" % (name.split("(")[0][:64], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
