import csv, glob, sys
t = glob.glob("gpurun_out/prof_mass/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(t)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-14:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    print("%-46s start %8.1f us  dur %8.2f us  grid %s" % (r["Kernel_Name"][:46], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", "")))
