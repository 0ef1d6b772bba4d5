#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
TAG=r04
python3 -m pytest tests/test_gpu_drivers.py tests/test_gpu_configs.py tests/test_mesh.py -x -q -m gpu > gpurun_out/r04_gputest_drivers.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r04_gputest_drivers.log
python3 bench.py --steps 20 --warmup 3 > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench_line.err; echo "bench rc=$?"
python3 bench.py --config c4 --steps 20 --warmup 3 > gpurun_out/${TAG}_bench_line_c4.json 2> gpurun_out/${TAG}_bench_line_c4.err; echo "bench c4 rc=$?"
python3 bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline --no-hbm-leg > gpurun_out/${TAG}_bench_line_c5.json 2> gpurun_out/${TAG}_bench_line_c5.err; echo "bench c5 rc=$?"
python3 tools/run_configs.py > gpurun_out/${TAG}_configs.txt 2>&1; echo "configs rc=$?"
python3 -c "
import json
for f in ('bench_line','bench_line_c4','bench_line_c5'):
    d=json.load(open('gpurun_out/r04_%s.json'%f)); r=d['roofline']
    print(f, d['value'], d['ms_per_step'], r.get('kernel_ms'), r.get('frac'), r.get('achieved'), (r.get('valu_issue') or {}).get('frac'), r.get('from_profile',{}).get('file'))
"
