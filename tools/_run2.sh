set -e
cd "$GRAFT_REPO_ROOT"
HU_INTERP_CULL=1 timeout -k 10 900 python -m pytest tests/test_gpu_culling.py tests/test_gpu_parity.py -x -q -m gpu -k "grid_eval or culled or sponge" > gpurun_out/cull_tests.log 2>&1 || { tail -40 gpurun_out/cull_tests.log; exit 1; }
tail -1 gpurun_out/cull_tests.log
for w in "sponge4 float4" "sponge3 float4" "sponge5 float4"; do
timeout -k 10 300 python tools/prof_cull.py $w 2>&1 | tail -1
HU_INTERP_CULL=1 timeout -k 10 300 python tools/prof_cull.py $w 2>&1 | tail -1
done
VARIANTS="default" bash tools/_run.sh
