#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/collect_kernels.sh into one JSON summary with a section per config and
kernel: calls, average / minimum duration (from --stats), the PMC counters per launch (averaged over the launches
of that kernel in the profiled run), per-wave figures, HBM traffic and the share of the VALU issue slots used."""
import collections
import csv
import glob
import json
import os
import re
import sys

root = sys.argv[1]
short = lambda name: name.split("(")[0].replace("void ", "").replace("sdfk::", "")
out = {}
for cfg in ("c3", "c5"):
    stats = {}
    for f in glob.glob(os.path.join(root, cfg + "_stats", "*", "*_kernel_stats.csv")):
        for r in csv.DictReader(open(f)):
            if "k_" in r["Name"]:
                stats[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]),
                                           "max_ns": float(r["MaxNs"]), "percent": float(r["Percentage"])}
    counters = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(root, cfg + "_pmc_*", "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if "k_" in r["Kernel_Name"]:
                counters[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    section = {}
    for name, st in stats.items():
        if not re.match(r"k_(grid_eval|classify)", name):
            continue
        c = {k: sum(v) / len(v) for k, v in counters.get(name, {}).items()}
        entry = dict(st, counters_per_launch=c)
        if "WRITE_SIZE" in c and "FETCH_SIZE" in c:
            # rocprofv3 reports KiB; gfx950 FETCH_SIZE undercounts wide coalesced reads by 2x (MI355X_MICROARCH.md "HBM")
            entry["hbm_traffic_bytes_per_launch"] = c["WRITE_SIZE"] * 1024 + 2 * c["FETCH_SIZE"] * 1024
        if c.get("SQ_WAVES"):
            entry["per_wave"] = {k: v / c["SQ_WAVES"] for k, v in c.items() if k.startswith("SQ_")}
        if "SQ_INSTS_VALU" in c:
            # a wave64 VALU instruction occupies its SIMD16 for 4 cycles; 1024 SIMDs; 2.4 GHz peak clock
            entry["valu_issue_busy"] = c["SQ_INSTS_VALU"] * 4 / (1024 * 2.4e9 * st["avg_ns"] * 1e-9)
        section[name] = entry
    out[cfg] = section
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
out["csrc_hash"] = bench.csrc_hash()
print(json.dumps(out, indent=1))
