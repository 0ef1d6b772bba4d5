#!/usr/bin/env python3
"""Timeline of a `rocprofv3 --kernel-trace` run of bench.py: per dense-kernel launch (A) the gap to the previous A, and what ran
between them -- the last 16 + 16 A launches are the timed direct steps and the timed graph replays."""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.split("(")[0].replace("void ", "").replace("sdfk::", "")[:48]
t0 = int(rows[0]["Start_Timestamp"])
a_idx = [i for i, r in enumerate(rows) if short(r["Kernel_Name"]).startswith("k_grid_eval<JitEval, 0, 2>")]
print("dense launches:", len(a_idx))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32
# in time order: ... timed direct steps (n), the graph's warm replays (3 x 8), the timed replays (n)
phases = {"direct": a_idx[-(2 * n + 24):-(n + 24)], "graph warm": a_idx[-(n + 24):-n], "graph": a_idx[-n:]}
for name, idx in phases.items():
    starts = [int(rows[i]["Start_Timestamp"]) for i in idx]
    ends = [int(rows[i]["End_Timestamp"]) for i in idx]
    print("== %s: A start-to-start (us): %s" % (name, " ".join("%.0f" % ((b - a) / 1e3) for a, b in zip(starts, starts[1:]))))
    print("   A durations (us): %s" % " ".join("%.0f" % ((e - s) / 1e3) for s, e in zip(starts, ends)))
    # one step in detail: everything between the 8th and 9th A
    if len(idx) < 10:
        continue
    lo, hi = idx[8], idx[9]
    base = int(rows[lo]["Start_Timestamp"])
    for r in rows[lo:hi + 1]:
        print("   %8.1f .. %8.1f us  q=%s %s" % ((int(r["Start_Timestamp"]) - base) / 1e3, (int(r["End_Timestamp"]) - base) / 1e3, r["Queue_Id"], short(r["Kernel_Name"])))
