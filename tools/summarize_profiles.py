#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (tools/collect_profiles.sh) into one JSON summary."""
import collections
import re
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
out = {"kernel_stats": [], "dense_kernel_counters_per_launch": {}}
for f in glob.glob(os.path.join(root, "bench_stats", "*", "*_kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        name = r["Name"]
        if "k_" in name or "rocclr" in name:
            out["kernel_stats"].append({"name": name.split("(")[0].replace("void ", "").replace("sdfk::", ""),
                                        "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                                        "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"]),
                                        "percent": float(r["Percentage"])})
for f in glob.glob(os.path.join(root, "pmc_*", "*", "*_counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if re.search(r"k_grid_eval<.*, 0, \d>", r["Kernel_Name"]):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out["dense_kernel_counters_per_launch"][k] = sum(v) / len(v)
c = out["dense_kernel_counters_per_launch"]
if "WRITE_SIZE" in c and "FETCH_SIZE" in c:
    # rocprofv3 reports KiB; gfx950 FETCH_SIZE undercounts wide coalesced reads by 2x
    # (MI355X_MICROARCH.md "HBM"): double it.  WRITE_SIZE is exact for 16-B/lane streaming stores.
    out["hbm_traffic_bytes_per_launch"] = c["WRITE_SIZE"] * 1024 + 2 * c["FETCH_SIZE"] * 1024
    out["hbm_traffic_note"] = "WRITE_SIZE*1024 + 2*FETCH_SIZE*1024 (gfx950 FETCH_SIZE halving correction)"
if "SQ_WAVES" in c:
    w = c["SQ_WAVES"]
    out["per_wave"] = {k: v / w for k, v in c.items() if k.startswith("SQ_")}
    dense = [k for k in out["kernel_stats"] if re.search(r"k_grid_eval<JitEval, 0, \d>|k_grid_eval<InterpEval<false>, 0, \d>", k["name"])]
    if dense and "SQ_INSTS_VALU" in c:
        # Share of the chip's VALU issue slots the kernel used: a wave64 VALU instruction occupies its SIMD16 for 4
        # cycles; 1024 SIMDs; 2.4 GHz peak clock; the kernel's average duration from the --stats pass.  (Round 1
        # divided SQ_ACTIVE_INST_VALU by GRBM_GUI_ACTIVE / 8 and got 1.025: that counter adds up per-wave
        # execution windows, which overlap in the pipeline, so it is not a utilisation.)
        avg_s = max(dense, key=lambda k: k["calls"])["avg_ns"] * 1e-9
        out["valu_issue_busy"] = c["SQ_INSTS_VALU"] * 4 / (1024 * 2.4e9 * avg_s)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
out["csrc_hash"] = bench.csrc_hash()   # the device code these counters describe (bench.py ignores a profile of other code)
print(json.dumps(out, indent=1))
