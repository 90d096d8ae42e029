cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3p; rm -rf $O; mkdir -p $O
python -m pytest tests/test_axial_layer_gpu.py -x -q 2>&1 | tail -2
python bench.py --roofline-only 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read())['roofline']; print('fwd ms', j['launch_ms'], 'bwd', j['bwd_core'])"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/roof -- python bench.py --roofline-only > $O/roof.log 2>&1
find $O -name "*kernel_trace.csv" -size +30M -delete
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r3p/roof/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(r['Name'][:80], r['Calls'], float(r['AverageNs'])/1000)
PY
