#!/usr/bin/env python3
"""Copy the judged pieces of gpurun_out/prof_<tag>/ (tools/collect_profiles.sh) into profiles/<prefix>_*:
the rocprofv3 kernel stats of the bench run, the bench line printed under rocprofv3, and the summary
(kernel averages + PMC counters), annotated.  Usage: publish_profiles.py <tag> <prefix> <evaluator> "<note>" """
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, prefix, evaluator, note = sys.argv[1:5]
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
stats = max(glob.glob(os.path.join(src, "bench_stats", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
shutil.copy(stats, os.path.join(dst, prefix + "_bench_kernel_stats.csv"))
line = [l for l in open(os.path.join(src, "bench_under_rocprof.log")) if l.startswith('{"metric"')][-1]
open(os.path.join(dst, prefix + "_bench_line_under_rocprof.json"), "w").write(line)
summary = json.load(open(os.path.join(src, "summary.json")))
summary.update({"grid_edge": 512, "round": int(prefix[1:3]) if prefix[1:3].isdigit() else 2, "config": "c3", "evaluator": evaluator, "note": note,
                "command": "tools/collect_profiles.sh: rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 2 "
                           "--no-cpu-baseline --no-hbm-leg ; PMC passes: rocprofv3 --pmc <counters> --kernel-trace -- python3 tools/prof_dense.py 512 3 "
                           "(CODECAD_AMD_SPECIALIZE=%s)" % ("1" if evaluator == "specialised" else "0")})
json.dump(summary, open(os.path.join(dst, prefix + "_summary.json"), "w"), indent=1)
print("published", prefix, "dense kernel avg %.3f ms" % (summary["kernel_stats"][0]["avg_ns"] / 1e6))
