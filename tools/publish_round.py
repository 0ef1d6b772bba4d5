#!/usr/bin/env python3
"""Copy the judged pieces of a tools/collect_round.sh run from gpurun_out/ into profiles/<tag>_*.
Usage: python tools/publish_round.py <tag>"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
src, dst = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
kernels = os.path.join(src, "prof_%s_kernels" % tag)
summary = json.load(open(os.path.join(kernels, "summary.json")))
summary["note"] = ("tools/collect_kernels.sh: rocprofv3 --kernel-trace --stats and separate --pmc passes over bench.py --config c3 / c4 / c5 "
                   "(--no-graph --no-hbm-leg --no-cpu-baseline), per kernel and launch; hbm_traffic = WRITE_SIZE*1024 + 2*FETCH_SIZE*1024 "
                   "(the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md); valu_issue_busy = SQ_INSTS_VALU * 4 / (1024 SIMDs * 2.4 GHz * avg duration)")
json.dump(summary, open(os.path.join(dst, "%s_kernels_summary.json" % tag), "w"), indent=1)
for cfg in ("c3", "c4", "c5"):
    stats = glob.glob(os.path.join(kernels, cfg + "_stats", "*", "*_kernel_stats.csv"))
    if stats:
        shutil.copy(max(stats, key=os.path.getmtime), os.path.join(dst, "%s_bench_%s_kernel_stats.csv" % (tag, cfg)))
consumers = os.path.join(src, "prof_%s_consumers" % tag)
for name in ("mesh", "render", "polygon"):
    stats = glob.glob(os.path.join(consumers, name, "*", "*_kernel_stats.csv"))
    if stats:
        shutil.copy(max(stats, key=os.path.getmtime), os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, name)))
for name in ("bench_line.json", "bench_line_forced_collectives.json", "bench_line_forced_replicated.json", "bench_line_c4.json", "bench_line_c4_forced_collectives.json", "bench_line_c5.json", "bench_line_c5_forced_collectives.json",
             "bench_line_interpreter.json", "hbm_sweep.jsonl", "configs.txt", "jit.txt"):
    path = os.path.join(src, "%s_%s" % (tag, name))
    if os.path.exists(path) and os.path.getsize(path):
        shutil.copy(path, os.path.join(dst, "%s_%s" % (tag, name)))
print("published", tag, "csrc_hash", summary.get("csrc_hash"))
