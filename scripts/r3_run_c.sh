cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3c; rm -rf $O; mkdir -p $O
python -m pytest tests/test_axial_layer_gpu.py -x -q 2>&1 | tail -5 > $O/tests.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/roof -- python bench.py --roofline-only > $O/roof.log 2>&1
python bench.py --no-cpu-baseline --no-roofline > $O/step.json 2>$O/step.err
MEDT_BWD_NW=1 python bench.py --roofline-only > $O/roof_nw1.json 2>/dev/null
MEDT_BWD_NW=4 python bench.py --roofline-only > $O/roof_nw4.json 2>/dev/null
MEDT_ROOF_AXIS=h python bench.py --roofline-only > $O/roof_h.json 2>/dev/null
find $O -name "*kernel_trace.csv" -size +30M -delete
tail -2 $O/tests.log; cat $O/step.json | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('step', j['ms_per_step'])"
for f in roof_nw1 roof_nw4 roof_h; do python -c "import json,sys; j=json.loads(open('$O/$f.json').read()); print('$f', j['roofline']['bwd_core']['launch_ms'])"; done
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r3c/roof/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(r['Name'][:80], r['Calls'], float(r['AverageNs'])/1000)
PY
