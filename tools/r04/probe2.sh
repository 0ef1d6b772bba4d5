#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04_probe2
python3 -m pytest tests/test_gpu_drivers.py -x -q -m gpu -k "full_size_512 or single_workgroup or library_forms" > gpurun_out/r04_probe2/pytest.log 2>&1; echo "pytest rc=$?"
python3 bench.py --steps 10 --warmup 2 > gpurun_out/r04_probe2/bench.json 2> gpurun_out/r04_probe2/bench.err; echo "bench rc=$?"
python3 bench.py --config c4 --steps 10 --warmup 2 > gpurun_out/r04_probe2/bench_c4.json 2> gpurun_out/r04_probe2/bench_c4.err; echo "bench c4 rc=$?"
CODECAD_AMD_FORCE_COLLECTIVES=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --config c4 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r04_probe2/bench_c4_forced.json 2> gpurun_out/r04_probe2/bench_c4_forced.err; echo "bench c4 forced rc=$?"
CFGS=c4 bash tools/collect_kernels.sh r04a > gpurun_out/r04_probe2/collect.log 2>&1; echo "collect rc=$?"
tail -3 gpurun_out/r04_probe2/pytest.log; cut -c1-600 gpurun_out/r04_probe2/bench_c4.json
