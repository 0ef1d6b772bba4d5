#!/bin/bash
# Run ON THE GPU BOX: planetary (BASELINE C4) under the plain and the deferred generator, with kernel stats.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04_planetary_probe
rm -rf "$OUT" && mkdir -p "$OUT"
python3 tools/prof_planetary.py > "$OUT/plain.txt" 2>&1; echo "plain rc=$?"
HU_MAX_PATHS=400 python3 tools/prof_planetary.py > "$OUT/deferred400.txt" 2>&1; echo "deferred rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/plain_stats" -- python3 tools/prof_planetary.py > "$OUT/plain_stats.log" 2>&1; echo "stats rc=$?"
HU_MAX_PATHS=400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/def_stats" -- python3 tools/prof_planetary.py > "$OUT/def_stats.log" 2>&1; echo "stats rc=$?"
cat "$OUT/plain.txt" "$OUT/deferred400.txt"
