#!/bin/bash
# Run ON THE GPU BOX (through gpurun): everything profiles/ holds for a round, in one call.
#   collect_kernels.sh      rocprofv3 --kernel-trace --stats + separate --pmc passes over whole bench steps (c3, c4 and c5),
#                           summarised per kernel (dense, leaf-block, classification; both evaluators' dense kernels)
#   collect_consumer_profiles.sh   kernel stats of the mesh pipeline, the renderers, contouring
#   the free-running bench lines: c3, c3 with forced collectives under torchrun, c5, c5 forced; the interpreter's line
#   prof_hbm.py (HBM-bound regime), run_configs.py (BASELINE configs through the library), prof_jit.py (compile cost, policy)
# tools/publish_round.py then copies the judged pieces into profiles/<tag>_*.
set -u
TAG=${1:-r04}
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
if [ "${PART:-all}" != "lines" ]; then
bash tools/collect_kernels.sh $TAG > gpurun_out/collect_${TAG}_kernels.log 2>&1; echo "kernels rc=$?"
bash tools/collect_consumer_profiles.sh ${TAG}_consumers > gpurun_out/collect_${TAG}_consumers.log 2>&1; echo "consumers rc=$?"
fi
if [ "${PART:-all}" = "kernels" ]; then exit 0; fi
python3 bench.py --steps 20 --warmup 3 > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench_line.err; echo "bench rc=$?"
# one rank in a real RCCL group: the default (the small levels replicated with ownership: no collective in this hierarchy) and
# with every level exchanged (all-gather + hu_slice_rows per level)
CODECAD_AMD_FORCE_COLLECTIVES=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 \
  bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --no-hbm-leg > gpurun_out/${TAG}_bench_line_forced_replicated.json 2> gpurun_out/${TAG}_bench_line_forced_replicated.err; echo "bench forced (replicated) rc=$?"
CODECAD_AMD_FORCE_COLLECTIVES=1 CODECAD_AMD_REPLICATE_SAMPLES=0 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29542 \
  bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --no-hbm-leg > gpurun_out/${TAG}_bench_line_forced_collectives.json 2> gpurun_out/${TAG}_bench_line_forced_collectives.err; echo "bench forced (exchanged) rc=$?"
python3 bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline --no-hbm-leg > gpurun_out/${TAG}_bench_line_c5.json 2> gpurun_out/${TAG}_bench_line_c5.err; echo "bench c5 rc=$?"
CODECAD_AMD_FORCE_COLLECTIVES=1 CODECAD_AMD_REPLICATE_SAMPLES=0 python3 bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline --no-hbm-leg > gpurun_out/${TAG}_bench_line_c5_forced_collectives.json 2> gpurun_out/${TAG}_bench_line_c5_forced_collectives.err; echo "bench c5 forced rc=$?"
python3 bench.py --config c4 --steps 10 --warmup 2 > gpurun_out/${TAG}_bench_line_c4.json 2> gpurun_out/${TAG}_bench_line_c4.err; echo "bench c4 rc=$?"
CODECAD_AMD_FORCE_COLLECTIVES=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29543 \
  bench.py --gpus 1 --config c4 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_bench_line_c4_forced_collectives.json 2> gpurun_out/${TAG}_bench_line_c4_forced_collectives.err; echo "bench c4 forced rc=$?"
python3 bench.py --evaluator interpreter --steps 10 --warmup 2 --no-cpu-baseline --no-hbm-leg > gpurun_out/${TAG}_bench_line_interpreter.json 2> gpurun_out/${TAG}_bench_line_interpreter.err; echo "bench interpreter rc=$?"
python3 tools/prof_hbm.py > gpurun_out/${TAG}_hbm_sweep.jsonl 2> gpurun_out/${TAG}_hbm_sweep.err; echo "hbm rc=$?"
python3 tools/run_configs.py > gpurun_out/${TAG}_configs.txt 2>&1; echo "configs rc=$?"
PROF_JIT_POOL=1 python3 tools/prof_jit.py > gpurun_out/${TAG}_jit.txt 2>&1; echo "jit rc=$?"
cut -c1-300 gpurun_out/${TAG}_bench_line.json
