#!/bin/bash
# Run ON THE GPU BOX (through gpurun): everything profiles/ holds for a round, in one go -- the PMC / kernel-stats
# profiles of both evaluators (collect_profiles.sh), the consumers' kernel stats (collect_consumer_profiles.sh), the
# HBM-regime sweep, the free-running bench lines (c3 and c5) and the BASELINE configs.  tools/publish_profiles.py and
# a few copies (profiles/README.md) then move the judged pieces into profiles/.
set -u
TAG=${1:-r02}
cd "$GRAFT_REPO_ROOT"
bash tools/collect_profiles.sh $TAG > gpurun_out/collect_$TAG.log 2>&1; echo "spec rc=$?"
CODECAD_AMD_SPECIALIZE=0 bash tools/collect_profiles.sh ${TAG}_interpreter > gpurun_out/collect_${TAG}_interpreter.log 2>&1; echo "interp rc=$?"
bash tools/collect_consumer_profiles.sh ${TAG}_consumers > gpurun_out/collect_${TAG}_consumers.log 2>&1; echo "consumers rc=$?"
python3 tools/prof_hbm.py > gpurun_out/${TAG}_hbm_sweep.jsonl 2> gpurun_out/${TAG}_hbm_sweep.err; echo "hbm rc=$?"
python3 bench.py --steps 20 --warmup 3 > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench_line.err; echo "bench rc=$?"
python3 bench.py --config c5 --steps 5 --warmup 2 > gpurun_out/${TAG}_bench_line_c5.json 2> gpurun_out/${TAG}_bench_line_c5.err; echo "bench c5 rc=$?"
python3 tools/run_configs.py > gpurun_out/${TAG}_configs.txt 2>&1; echo "configs rc=$?"
tail -1 gpurun_out/${TAG}_bench_line.json | cut -c1-400
