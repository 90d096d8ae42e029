cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3e; rm -rf $O; mkdir -p $O
python -m pytest tests/test_axial_layer_gpu.py -x -q 2>&1 | tail -4 > $O/tests.log; tail -2 $O/tests.log
for ls in 16 32; do
  MEDT_BWD_LS=$ls python -m pytest tests/test_axial_layer_gpu.py -x -q 2>&1 | tail -2 > $O/tests_ls$ls.log
  MEDT_BWD_LS=$ls python bench.py --roofline-only > $O/roof_ls$ls.json 2>/dev/null
  MEDT_BWD_LS=$ls python bench.py --no-cpu-baseline --no-roofline > $O/step_ls$ls.json 2>/dev/null
  echo "LS=$ls: $(tail -1 $O/tests_ls$ls.log)"
  python -c "import json; j=json.loads(open('$O/roof_ls$ls.json').read()); print(' bwd_core ms', j['roofline']['bwd_core']['launch_ms'])"
  python -c "import json; j=json.loads(open('$O/step_ls$ls.json').read()); print(' step ms', j['ms_per_step'])"
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/roof -- python bench.py --roofline-only > $O/roof.log 2>&1
find $O -name "*kernel_trace.csv" -size +30M -delete
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r3e/roof/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:7]:
    print(r['Name'][:80], r['Calls'], float(r['AverageNs'])/1000)
PY
