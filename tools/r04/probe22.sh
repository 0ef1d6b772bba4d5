#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_probe22; mkdir -p $O
run() { # name, env...
  name=$1; shift
  env "$@" python3 bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-hbm-leg --no-graph > $O/$name.json 2> $O/$name.err; echo "$name rc=$?"
}
run base BENCH_OVERLAP_LEAF=0
run ov_pad0 BENCH_OVERLAP_LEAF=1
for pad in 8 16 24 32; do
  run ov_pad$pad BENCH_OVERLAP_LEAF=1 HU_DENSE_BOX_PAD=$pad
  run noov_pad$pad BENCH_OVERLAP_LEAF=0 HU_DENSE_BOX_PAD=$pad
done
run base2 BENCH_OVERLAP_LEAF=0
python3 -c "
import json,glob
for f in ['base','ov_pad0','ov_pad8','noov_pad8','ov_pad16','noov_pad16','ov_pad24','noov_pad24','ov_pad32','noov_pad32','base2']:
    try:
        d=json.load(open('$O/%s.json'%f)); r=d['roofline']
        print('%-12s'%f, d['value'], d['ms_per_step'], 'A', r.get('kernel_ms'), 'C', d.get('adaptive',{}).get('leaf_blocks_ms'), d.get('verified',{}).get('ok'))
    except Exception as e: print(f, 'failed', e)
"
