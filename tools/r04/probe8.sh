#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_probe8; rm -rf $O; mkdir -p $O
for w in 0 2 4 6; do
  if [ $w = 0 ]; then F=""; else F="-DSDF_WAVES_PER_EU=$w"; fi
  HU_RTC_FLAGS="$F" python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-verify --no-graph > $O/bench_w$w.json 2> $O/bench_w$w.err; python3 -c "
import json
d=json.load(open('$O/bench_w$w.json'))
print('waves/EU $w: step', d['ms_per_step'], 'dense', d['roofline']['kernel_ms'], 'leaf', d['adaptive']['leaf_blocks_ms'])
for e in d.get('roofline_hbm',[]):
    if e['tape'] in ('box','sphere'): print('    ', e['tape'], e['kernel'], e['ms'], e['frac'])
"; done
