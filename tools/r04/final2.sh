#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_variants.py tests/test_gpu_math.py tests/test_gpu_parity.py tests/test_gpu_random_shapes.py tests/test_gpu_shape_equality.py tests/test_mesh.py tests/test_polygon2d_render.py tests/test_render_baselines.py -x -q -m gpu > gpurun_out/r04_gputest_rest.log 2>&1; echo "pytest rest rc=$?"; tail -4 gpurun_out/r04_gputest_rest.log
PART=lines bash tools/collect_round.sh r04
