#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_probe24; mkdir -p $O
export CODECAD_AMD_CACHE=0
for v in 24 16 12 8 4; do
  HU_TAB_PAIR_TOTAL=$v python3 bench.py --config c4 --steps 20 --warmup 3 --no-cpu-baseline > $O/c4_$v.json 2> $O/c4_$v.err; echo "c4 $v rc=$?"
  HU_TAB_PAIR_TOTAL=$v python3 tools/prof_planetary.py > $O/plan_$v.txt 2>&1
  HU_TAB_PAIR_TOTAL=$v python3 bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline --no-hbm-leg --no-graph > $O/c5_$v.json 2> $O/c5_$v.err; echo "c5 $v rc=$?"
done
python3 -c "
import json
for c in ('c4','c5'):
  for v in (24,16,12,8,4):
    try:
        d=json.load(open('$O/%s_%d.json'%(c,v))); r=d['roofline']
        print(c, 'pair_total', v, d['value'], d['ms_per_step'], r.get('kernel_ms'), d.get('verified',{}).get('ok'))
    except Exception as e: print(c, v, 'failed', e)
"
for v in 24 16 12 8 4; do echo "pair_total $v: $(grep per-tape $O/plan_$v.txt | cut -c1-200)"; done
