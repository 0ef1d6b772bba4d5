#!/bin/bash
# Run ON THE GPU BOX: per-kernel hipRTC builds with the precompiled header -- tests, build latencies, and the bench's kernels
# compiled by this process's hipRTC (the torch wheel's) against the compile servers' (the ROCm installation's).
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_probe13; mkdir -p $O
ls -la codecad_amd/hip_util/pch/ > $O/pch_ls.txt 2>&1
python3 -m pytest tests/test_gpu_variants.py tests/test_gpu_pruning.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
PROF_JIT_POOL=1 python3 tools/prof_jit.py > $O/jit.txt 2>&1; echo "jit rc=$?"
HU_RTC_PCH=0 python3 tools/prof_jit.py > $O/jit_nopch.txt 2>&1; echo "jit nopch rc=$?"
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
CODECAD_AMD_SPECIALIZE_POOL=1 CODECAD_AMD_CACHE=0 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_pool.json 2> $O/bench_pool.err; echo "bench pool rc=$?"
python3 bench.py --config c4 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err; echo "bench c4 rc=$?"
CODECAD_AMD_SPECIALIZE_POOL=1 CODECAD_AMD_CACHE=0 python3 bench.py --config c4 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_c4_pool.json 2> $O/bench_c4_pool.err; echo "bench c4 pool rc=$?"
grep -v amdgpu.ids $O/jit.txt; echo; grep -v amdgpu.ids $O/jit_nopch.txt | tail -3
python3 -c "
import json
for f in ('bench','bench_pool','bench_c4','bench_c4_pool'):
    try:
        d=json.load(open('$O/%s.json'%f)); r=d['roofline']
        print(f, d['value'], d['ms_per_step'], r.get('kernel_ms'), r.get('frac'), d.get('verified',{}).get('ok'), d.get('adaptive',{}).get('leaf_blocks_ms'))
    except Exception as e: print(f, 'failed', e)
"
