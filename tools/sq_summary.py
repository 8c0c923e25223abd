"""Per-kernel SQ counter summary from one rocprofv3 --pmc pass (tools/gpu_sqpmc.sh): where the wave cycles go.
Per the MI355X guide: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles and
WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~= WAVE_CYCLES; SQ_VALU_MFMA_BUSY_CYCLES counts cycles;
SQ_LDS_BANK_CONFLICT = extra LDS cycles out of SQ_LDS_IDX_ACTIVE."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            key = (name[:60], int(r["Grid_Size"]))
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for (name, grid), cs in sorted(acc.items()):
        m = {k: sum(v) / len(v) for k, v in cs.items()}
        wc = m.get("SQ_WAVE_CYCLES", 0.0)
        if wc <= 0 or not (name.startswith("k_") or name.startswith("k_conv") or "k_" in name):
            continue
        busy = m.get("SQ_BUSY_CYCLES", 0.0)
        row = {"grid": grid, "launches": len(cs.get("SQ_WAVE_CYCLES", [])),
               "wait_any_frac": round(m.get("SQ_WAIT_ANY", 0.0) / wc, 3),
               "wait_inst_frac": round(m.get("SQ_WAIT_INST_ANY", 0.0) / wc, 3),
               "active_inst_frac": round(m.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 3),
               "lds_conflict_frac_of_lds_active": round(m.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(1.0, m.get("SQ_LDS_IDX_ACTIVE", 0.0)), 3),
               "mfma_busy_cycles": round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)),
               "sq_busy_cycles": round(busy),
               # MFMA-busy as a fraction of the launch: SQ_VALU_MFMA_BUSY_CYCLES sums the 1024 SIMDs' busy cycles, SQ_BUSY_CYCLES
               # the 32 shader engines' (a 12.5 us launch = 30 k cycles reads 0.84 M): busy / (32 x SQ_BUSY) = mean over the SIMDs
               "mfma_busy_frac": round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(1.0, 32.0 * busy), 3)}
        out["%s [grid %d]" % (name, grid)] = row
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
