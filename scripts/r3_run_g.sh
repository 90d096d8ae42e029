cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3g; rm -rf $O; mkdir -p $O
python -m pytest tests/test_axial_layer_gpu.py -x -q 2>&1 | tail -4 > $O/tests.log; tail -2 $O/tests.log
python bench.py --no-cpu-baseline --no-roofline > $O/step.json 2>/dev/null
python -c "import json; j=json.loads(open('$O/step.json').read()); print(' step ms', j['ms_per_step'])"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/roof -- python bench.py --roofline-only > $O/roof.log 2>&1
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc -- python bench.py --roofline-only > $O/pmc.log 2>&1
find $O -name "*kernel_trace.csv" -size +30M -delete
python - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/r3g/roof/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:7]:
    print(r['Name'][:80], r['Calls'], float(r['AverageNs'])/1000)
acc=collections.defaultdict(list)
for f in glob.glob('gpurun_out/r3g/pmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'sweep' in r['Kernel_Name'] and 'false' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
print({k: sum(v)/len(v) for k,v in acc.items()})
PY
