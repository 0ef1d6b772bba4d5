set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in $VARIANTS; do
L=$GRAFT_REPO_ROOT/build/variants/$v.so; [ $v = default ] && L=$GRAFT_REPO_ROOT/codecad_amd/hip_util/libhip_util.so
echo $v
CODECAD_AMD_LIB=$L timeout -k 10 300 python tools/prof_cull.py sponge4 float4 2>&1 | tail -1
CODECAD_AMD_LIB=$L timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH --output-format csv -d gpurun_out/pmc_$v -- python3 tools/prof_cull.py sponge4 float4 > /dev/null 2>&1
python - <<PY
import csv,glob,collections
acc=collections.Counter(); n=collections.Counter()
for f in glob.glob('gpurun_out/pmc_$v/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_grid_eval' in r['Kernel_Name']:
            acc[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
w=acc['SQ_WAVES']/n['SQ_WAVES']
print({k: round(acc[k]/n[k]/w,1) for k in acc})
PY
done
