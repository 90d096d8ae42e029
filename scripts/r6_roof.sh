#!/bin/bash
# Round 6: the roofline leg with the steady-state timing + rocprofv3 --kernel-trace --stats of the SAME command (the averages must agree with launch_ms
# up to the share of warm-up launches in them), then the default bench line.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6_roof; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python bench.py --roofline-only > $O/roofline_only.json 2> $O/roofline_only.err
S=$(ls -S $(find $O/p -name "*kernel_stats.csv") | head -1); cp $S $O/roofline_kernel_stats.csv; rm -rf $O/p
python - $O <<'PY'
import json, sys, csv
O = sys.argv[1]
j = json.loads([l for l in open(O + "/roofline_only.json") if l.startswith("{")][-1])
r = j["roofline"]
print("roofline-only: launch_ms", r["launch_ms"], "cold", r.get("launch_ms_cold_burst"), "frac", r["frac"], "bwd_core", r["bwd_core"]["launch_ms"], r["bwd_core"]["frac"], "stats", r["stats_kernel"]["launch_ms"])
a = j["also"]; print("also: launch_ms", a["launch_ms"], "frac", a["frac"], "bwd_core", a["bwd_core"]["launch_ms"], a["bwd_core"]["frac"])
for row in csv.DictReader(open(O + "/roofline_kernel_stats.csv")):
    n = row["Name"]
    if any(k in n for k in ("attn_fwd4r", "attn_fwd3", "sim_stats_rows", "attn_bwd_sweep", "attn_bwd_fix", "attn_bwd_relfix")):
        print("  rocprof %-70s calls %5s avg %8.1f us min %8.1f max %8.1f" % (n.split("(")[0][-70:], row["Calls"], float(row["AverageNs"]) / 1e3, float(row["MinNs"]) / 1e3, float(row["MaxNs"]) / 1e3))
PY
timeout 600 python bench.py 2>$O/bench.err | tail -1 > $O/bench_line.json
python -c "import json; j=json.load(open('$O/bench_line.json')); r=j['roofline']; print('bench: step', j['ms_per_step'], j['value'], 'roof launch_ms', r['launch_ms'], 'cold', r.get('launch_ms_cold_burst'), 'frac', r['frac'], 'bwd', r['bwd_core']['frac'], 'also', [(a['frac'], a['bwd_core']['frac']) for a in r['also']], r['in_model_shape'])"
