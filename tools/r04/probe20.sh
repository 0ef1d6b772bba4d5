#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_probe20; mkdir -p $O
export CODECAD_AMD_CACHE=0
for rep in 1 2; do
python3 bench.py --config c4 --steps 20 --warmup 3 --no-cpu-baseline > $O/c4_base_$rep.json 2> $O/c4_base_$rep.err; echo "base rc=$?"
HU_RTC_FLAGS="-DSDF_CLASSIFY_ORDER=1" python3 bench.py --config c4 --steps 20 --warmup 3 --no-cpu-baseline > $O/c4_order_$rep.json 2> $O/c4_order_$rep.err; echo "order rc=$?"
done
HU_RTC_FLAGS="-DSDF_CLASSIFY_ORDER=1" python3 -m pytest tests/test_gpu_pruning.py tests/test_gpu_configs.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
python3 -c "
import json
for f in ('c4_base_1','c4_order_1','c4_base_2','c4_order_2'):
    try:
        d=json.load(open('$O/%s.json'%f)); r=d['roofline']
        print(f, d['value'], d['ms_per_step'], r.get('kernel_ms'), d.get('verified',{}).get('ok'))
    except Exception as e: print(f, 'failed', e)
"
