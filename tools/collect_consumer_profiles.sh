#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel stats of the consumers of the path -- the mesh pipeline
# (subdivision, leaf-block evaluation, marching cubes, STL records) and the renderers -- per-tape code.
# Outputs land in gpurun_out/prof_<tag>/; tools/publish_profiles.py does not handle these: copy the two
# *_kernel_stats.csv into profiles/ by hand (see profiles/README.md).
set -u
TAG=${1:-r01_consumers}
export CODECAD_AMD_SPECIALIZE=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT" && mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/mesh" -- python3 tools/prof_mesh.py > "$OUT/mesh.log" 2>&1
echo "mesh rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/render" -- python3 tools/prof_render.py > "$OUT/render.log" 2>&1
echo "render rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/polygon" -- python3 tools/prof_polygon.py > "$OUT/polygon.log" 2>&1
echo "polygon rc=$?"
