set -e
cd "$GRAFT_REPO_ROOT"
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -1 gpurun_out/gpu_tests.log
for w in "sponge4 float4" "sponge4 float" "sponge5 float4" "sponge3 float4" "csg float4"; do
timeout -k 10 300 python tools/prof_cull.py $w 2>&1 | tail -1
done
python tools/run_configs.py 2>&1 | grep -v amdgpu.ids | head -12
