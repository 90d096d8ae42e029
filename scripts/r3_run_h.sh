cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3h; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/step -- python bench.py --no-cpu-baseline --no-roofline > $O/step.log 2>&1
find $O -name "*kernel_trace.csv" -size +30M -delete
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r3h/step/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print(r['Name'][:90], r['Calls'], round(float(r['TotalDurationNs'])/1000), round(float(r['AverageNs'])/1000,1))
PY
tail -1 $O/step.log | cut -c1-200
