#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_probe25; mkdir -p $O
export CODECAD_AMD_CACHE=0 AMD_COMGR_CACHE=0
for opt in O3 O1; do
  F=""; [ "$opt" = "O1" ] && F="-O1"
  HU_RTC_FLAGS="$F" python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-hbm-leg --no-graph > $O/c3_$opt.json 2> $O/c3_$opt.err; echo "c3 $opt rc=$?"
  HU_RTC_FLAGS="$F" python3 bench.py --config c4 --steps 20 --warmup 3 --no-cpu-baseline > $O/c4_$opt.json 2> $O/c4_$opt.err; echo "c4 $opt rc=$?"
  HU_RTC_FLAGS="$F" python3 bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline --no-hbm-leg --no-graph > $O/c5_$opt.json 2> $O/c5_$opt.err; echo "c5 $opt rc=$?"
  HU_RTC_FLAGS="$F" python3 tools/prof_planetary.py > $O/plan_$opt.txt 2>&1
  HU_RTC_FLAGS="$F" python3 tools/prof_jit.py > $O/jit_$opt.txt 2>&1
done
HU_RTC_FLAGS="-O1" python3 -m pytest tests/test_gpu_bricks.py tests/test_gpu_pruning.py tests/test_gpu_variants.py -x -q -m gpu > $O/pytest_O1.log 2>&1; echo "pytest O1 rc=$?"; tail -2 $O/pytest_O1.log
python3 -c "
import json
for c in ('c3','c4','c5'):
  for v in ('O3','O1'):
    try:
        d=json.load(open('$O/%s_%s.json'%(c,v))); r=d['roofline']
        print(c, v, d['value'], d['ms_per_step'], r.get('kernel_ms'), d.get('verified',{}).get('ok'), d.get('adaptive',{}).get('leaf_blocks_ms'))
    except Exception as e: print(c, v, 'failed', e)
"
for v in O3 O1; do echo "$v: $(grep per-tape $O/plan_$v.txt | cut -c1-200)"; grep "default policy" $O/jit_$v.txt | cut -c1-220; done
