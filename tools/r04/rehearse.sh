#!/bin/bash
# Run ON THE GPU BOX: the multi-rank paths with several gloo ranks sharing ONE MI355X (host-staged collectives): functional
# rehearsal of replicated levels with ownership, the exchanged path and the bench at N = 2, 3, 4 -- not a measurement.
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_rehearsal.txt; : > $O
export CODECAD_AMD_DIST_BACKEND=gloo
for n in 2 3; do
  echo "# tools/rehearse_dist.py, $n gloo ranks on one MI355X" >> $O
  python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2951$n tools/rehearse_dist.py >> $O 2>&1; echo "rehearse $n rc=$?"
done
for n in 2 4; do
  echo "# bench.py --gpus $n (replicated levels with ownership), gloo ranks on one MI355X" >> $O
  python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2952$n bench.py --gpus $n --steps 4 --warmup 1 >> $O 2>&1; echo "bench $n rc=$?"
done
echo "# bench.py --gpus 2 with every level exchanged (CODECAD_AMD_REPLICATE_SAMPLES=0)" >> $O
CODECAD_AMD_REPLICATE_SAMPLES=0 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --steps 4 --warmup 1 >> $O 2>&1; echo "bench 2 exchanged rc=$?"
echo "# bench.py --gpus 3 --config c4" >> $O
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29532 bench.py --gpus 3 --config c4 --steps 4 --warmup 1 >> $O 2>&1; echo "bench c4 3 rc=$?"
echo "# bench.py --gpus 2 --config c5" >> $O
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --config c5 --steps 2 --warmup 1 >> $O 2>&1; echo "bench c5 2 rc=$?"
grep -v "amdgpu.ids\|^$" $O | cut -c1-420
