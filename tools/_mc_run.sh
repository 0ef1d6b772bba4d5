set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_mesh.py -x -q -m gpu 2>&1 | tail -3
export CODECAD_AMD_SPECIALIZE=1
rm -rf gpurun_out/prof_mc && mkdir -p gpurun_out/prof_mc
for v in 0 2 3; do
L=$GRAFT_REPO_ROOT/build/variants/mc$v.so; [ $v = 0 ] && L=$GRAFT_REPO_ROOT/codecad_amd/hip_util/libhip_util.so
CODECAD_AMD_LIB=$L timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_mc/v$v -- python3 tools/prof_mesh.py > gpurun_out/prof_mc/v$v.log 2>&1
done
python - <<'PY'
import csv,glob
for v in (0,2,3):
  for f in glob.glob('gpurun_out/prof_mc/v%d/**/*kernel_stats.csv'%v, recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_mc_block' in r['Name']: print(v, r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e6)
PY
