#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/collect_kernels.sh into one JSON summary with a section per config and
kernel: calls, average / minimum / maximum duration (per dispatch, from the kernel trace of the stats pass), the PMC
counters per launch (averaged over the profiled launches of that kernel), per-wave figures, HBM traffic, registers and
the share of the VALU issue slots used.  A kernel that a step launches at very different sizes (c4: the 245-cell top
level and the 57.7 M-sample leaf level of k_classify) is summarised over its largest FREQUENT launch size only (`grid`)."""
import collections
import csv
import glob
import json
import os
import re
import sys

root = sys.argv[1]
short = lambda name: name.split("(")[0].replace("void ", "").replace("sdfk::", "")
wanted = lambda name: re.match(r"k_(grid_eval|classify|box_masks)", name) is not None


def waves_per_simd(vgprs):
    """MI355X_MICROARCH.md "Register files": 512 registers per lane per SIMD, allocated in granules of 8."""
    alloc = -(-int(vgprs) // 8) * 8
    return min(8, 512 // max(alloc, 8))


out = {}
for cfg in ("c3", "c4", "c5"):
    traces = glob.glob(os.path.join(root, cfg + "_stats", "*", "*_kernel_trace.csv"))
    if not traces:
        continue
    launches = collections.defaultdict(list)
    for f in traces:
        for r in csv.DictReader(open(f)):
            name = short(r["Kernel_Name"])
            if wanted(name):
                grid = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y") or 1) * int(r.get("Grid_Size_Z") or 1)
                launches[name].append({"grid": grid, "ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
                                       "vgprs": int(r["VGPR_Count"]) + int(r.get("Accum_VGPR_Count") or 0), "sgprs": int(r["SGPR_Count"])})
    counters = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(root, cfg + "_pmc_*", "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            name = short(r["Kernel_Name"])
            if wanted(name):
                counters[name][r["Counter_Name"]].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    section = {}
    total_ns = sum(l["ns"] for ls in launches.values() for l in ls) or 1
    for name, ls in launches.items():
        # the launch size the timed steps use: the largest among the FREQUENT sizes (a warm-up launch with a first-guess
        # list capacity is larger than any later one, and alone)
        freq = collections.Counter(l["grid"] for l in ls)
        grid = max(g for g, n in freq.items() if 2 * n >= max(freq.values()))
        big = [l for l in ls if l["grid"] == grid]
        ns = [l["ns"] for l in big]
        c = {}
        for k, v in counters.get(name, {}).items():
            vals = [x for g, x in v if g == grid]
            if vals:
                c[k] = sum(vals) / len(vals)
        entry = {"calls": len(big), "calls_of_any_size": len(ls), "grid": grid, "avg_ns": sum(ns) / len(ns), "min_ns": min(ns), "max_ns": max(ns),
                 "percent": 100.0 * sum(l["ns"] for l in ls) / total_ns, "vgprs": big[0]["vgprs"], "sgprs": big[0]["sgprs"],
                 "waves_per_simd_by_vgprs": waves_per_simd(big[0]["vgprs"]), "counters_per_launch": c}
        if "WRITE_SIZE" in c and "FETCH_SIZE" in c:
            # rocprofv3 reports KiB; gfx950 FETCH_SIZE undercounts wide coalesced reads by 2x (MI355X_MICROARCH.md "HBM")
            entry["hbm_traffic_bytes_per_launch"] = c["WRITE_SIZE"] * 1024 + 2 * c["FETCH_SIZE"] * 1024
        if c.get("SQ_WAVES"):
            entry["per_wave"] = {k: v / c["SQ_WAVES"] for k, v in c.items() if k.startswith("SQ_")}
        if "SQ_INSTS_VALU" in c:
            # a wave64 VALU instruction occupies its SIMD16 for 4 cycles; 1024 SIMDs; 2.4 GHz peak clock
            entry["valu_issue_busy"] = c["SQ_INSTS_VALU"] * 4 / (1024 * 2.4e9 * entry["avg_ns"] * 1e-9)
        section[name] = entry
    out[cfg] = section
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
out["csrc_hash"] = bench.csrc_hash()
print(json.dumps(out, indent=1))
