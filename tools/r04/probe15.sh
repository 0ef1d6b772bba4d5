#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_probe15; mkdir -p $O
export CODECAD_AMD_CACHE=0 CODECAD_AMD_SPECIALIZE_POOL=0
for rep in 1 2; do
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-hbm-leg > $O/base_$rep.json 2> $O/base_$rep.err; echo "base rc=$?"
HU_RTC_FLAGS="-DSDF_XCD_BOX_ORDER=1" python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-hbm-leg > $O/xcd_$rep.json 2> $O/xcd_$rep.err; echo "xcd rc=$?"
done
python3 -c "
import json
for f in ('base_1','xcd_1','base_2','xcd_2'):
    try:
        d=json.load(open('$O/%s.json'%f)); r=d['roofline']
        print(f, d['value'], d['ms_per_step'], r.get('kernel_ms'), r.get('frac'), d.get('verified',{}).get('ok'), d.get('adaptive',{}).get('leaf_blocks_ms'))
    except Exception as e: print(f, 'failed', e)
"
