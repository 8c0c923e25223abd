"""Per-launch HBM bytes of the candidate kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).
Units/corrections per MI355X_MICROARCH.md §HBM: both counters are in KiB; on gfx950 FETCH_SIZE reports half of
the bytes of a wide coalesced streaming read (128-B requests tallied at 64 B), so the read side is doubled.
usage: pmc_summary.py <fetch dir> <write dir>  ->  JSON keyed by bench.py's profiling tags."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load(d, counter):
    out = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            key = (r["Kernel_Name"], int(r["Grid_Size"]))
            out[key].append(float(r["Counter_Value"]))
    return out


def tag_of(name, grid, biggest):
    """bench.py tag of a kernel instance; the hidden layer's launches are the larger grid of k_nl_fwd2 / k_nl_bwd."""
    if "k_clip_adam" in name:
        return "clip_adam"
    if "k_fc_gemm_fwd" in name:            # the hidden layer as tiled GEMMs (fc_gemm.h, from 128 rows per net on)
        return "fc_h_fwd"
    if "k_fc_gemm_bwd" in name:
        return "fc_h_bwd"
    if "k_nl_fwd" in name:
        return "fc_h_fwd" if grid == biggest["k_nl_fwd"] and not biggest.get("gemm") else "fc_z_fwd"
    if "k_nl_bwd" in name:
        # the output layer's backward is the TALL instantiation at batch <= 32 (k_nl_bwd<true>); with the hidden layer on the tiled
        # GEMMs (batch >= 128) k_nl_bwd<false> is the output layer's.  (Not by grid size: the data-efficient net's output layer
        # has the larger grid, and the hidden layer's launch loses one workgroup when the early draw takes the write-back away.)
        if "k_nl_bwd<true>" in name:
            return "fc_z_bwd"
        return "fc_z_bwd" if biggest.get("gemm") else "fc_h_bwd"
    if "k_conv_dw_all" in name:
        return "conv_dw_all"
    if "k_sample" in name:
        return "sample"
    if "k_head<" in name:
        return "head"
    for kind, suffix in (("k_conv_fwd_lds", "fwd"), ("k_conv_fwd_t16", "fwd"), ("k_conv_fwd_multi", "fwd"), ("k_conv_fwd_full", "fwd"),
                         ("k_conv_dx_lds", "dx")):
        if kind in name:
            geo = name.split("ConvGeom<")[1].split(">")[0].replace(" ", "")
            layer = {"8,4,84,20": 1, "4,2,20,9": 2, "3,1,9,7": 3, "5,5,84,16": 1, "5,5,16,3": 2}.get(geo)
            return "conv%d_%s" % (layer, suffix) if layer else None
    return None


def main():
    fetch = load(sys.argv[1], "FETCH_SIZE")
    write = load(sys.argv[2], "WRITE_SIZE")
    biggest = {}
    for (name, grid) in fetch:
        for k in ("k_nl_fwd", "k_nl_bwd"):
            if k in name:
                biggest[k] = max(biggest.get(k, 0), grid)
        if "k_fc_gemm" in name:
            biggest["gemm"] = 1
    res = {}
    for key in sorted(fetch):
        tag = tag_of(key[0], key[1], biggest)
        if tag is None or len(fetch[key]) < 5:
            continue
        if tag in res and (res[tag]["launches"] > len(fetch[key]) if tag.startswith("fc_") else res[tag]["grid_size"] > key[1]):
            continue      # e.g. "sample": the learn step's launch (it hosts the optimiser pass) over the PER-only phase's;
                          # fc_*: the steady-state variant of the launch (most launches)
        f = sum(fetch[key]) / len(fetch[key])
        w = sum(write.get(key, [0])) / max(1, len(write.get(key, [0])))
        res[tag] = {"kernel": key[0].split("(")[0][:80], "grid_size": key[1], "launches": len(fetch[key]), "FETCH_SIZE_KiB": f,
                    "WRITE_SIZE_KiB": w, "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0,
                    "note": "FETCH_SIZE doubled (gfx950 half-count of wide coalesced reads), WRITE_SIZE uncalibrated"}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
