set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_mesh.py -x -q -m gpu 2>&1 | tail -3
export CODECAD_AMD_SPECIALIZE=1
rm -rf gpurun_out/prof_mc && mkdir -p gpurun_out/prof_mc
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_mc/v0 -- python3 tools/prof_mesh.py > gpurun_out/prof_mc/v0.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d gpurun_out/prof_mc/pmc1 -- python3 tools/prof_mesh.py > gpurun_out/prof_mc/pmc1.log 2>&1 || echo "pmc failed"
python - <<'PY'
import csv,glob,collections
for f in glob.glob('gpurun_out/prof_mc/v0/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_mc_' in r['Name']: print(r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e6)
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for f in glob.glob('gpurun_out/prof_mc/pmc*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:40]
        if 'k_mc_block' not in k: continue
        acc[k][r['Counter_Name']]+=float(r['Counter_Value']); n[(k,r['Counter_Name'])]+=1
for k in acc:
    w=acc[k]['SQ_WAVES']/n[(k,'SQ_WAVES')]
    print(k, {c: round(v/n[(k,c)]/w,1) for c,v in acc[k].items()})
PY
grep "mc kernels" gpurun_out/prof_mc/v0.log | tail -1
