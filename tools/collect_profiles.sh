#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 summaries for the bench and PMC passes for the
# dominant kernel.  Outputs land in gpurun_out/prof_<tag>/; copy what should be judged into profiles/.
# Counters are collected in their own passes, only with --kernel-trace (never with sys/hip traces).
set -u
TAG=${1:-r01}
# evaluator for the PMC passes: CODECAD_AMD_SPECIALIZE=1 (default, what bench.py measures) or 0
export CODECAD_AMD_SPECIALIZE=${CODECAD_AMD_SPECIALIZE:-1}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT" && mkdir -p "$OUT"
EVAL=auto; [ "$CODECAD_AMD_SPECIALIZE" = "0" ] && EVAL=interpreter
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_stats" -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-hbm-leg --evaluator $EVAL > "$OUT/bench_under_rocprof.log" 2>&1
echo "bench_stats rc=$?"
pass() { # name counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -- python3 tools/prof_dense.py 512 3 > "$OUT/$name.log" 2>&1
  echo "$name rc=$?"
}
pass pmc_insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass pmc_stalls SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
pass pmc_write WRITE_SIZE
pass pmc_fetch FETCH_SIZE
pass pmc_grbm GRBM_GUI_ACTIVE GRBM_COUNT
python3 tools/summarize_profiles.py "$OUT" > "$OUT/summary.json"
cat "$OUT/summary.json"
