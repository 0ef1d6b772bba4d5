#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 PMC passes over a whole bench step, summarised PER KERNEL -- the
# dense kernel, the leaf-block kernel (k_grid_eval_blocks) and the classification kernel (k_classify) -- for the c3
# step and for configs c4 (planetary mass properties) and c5.  Counters in their own passes, only with --kernel-trace.  -> gpurun_out/prof_<tag>_kernels/
set -u
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_${TAG}_kernels
rm -rf "$OUT" && mkdir -p "$OUT"
for CFG in ${CFGS:-c3 c4 c5}; do
  STEPS=4; [ "$CFG" = "c5" ] && STEPS=2
  ARGS="--config $CFG --steps $STEPS --warmup 1 --no-cpu-baseline --no-hbm-leg --no-graph --no-verify"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${CFG}_stats" -- python3 bench.py $ARGS > "$OUT/${CFG}_stats.log" 2>&1
  echo "$CFG stats rc=$?"
  pass() { # name counters...
    local name=$1; shift
    rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/${CFG}_$name" -- python3 bench.py $ARGS > "$OUT/${CFG}_$name.log" 2>&1
    echo "$CFG $name rc=$?"
  }
  pass pmc_insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES
  pass pmc_stalls SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD
  pass pmc_write WRITE_SIZE
  pass pmc_fetch FETCH_SIZE
  pass pmc_icache SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH
done
python3 tools/summarize_kernels.py "$OUT" > "$OUT/summary.json"
cat "$OUT/summary.json"
