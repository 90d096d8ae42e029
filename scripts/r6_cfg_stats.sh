#!/bin/bash
# Round 6: per-kernel-instance statistics of the replayed step of another BASELINE configuration (rocprofv3 --kernel-trace --stats).
# usage: r6_cfg_stats.sh <name> <bench.py arguments>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
name=$1; shift
O=gpurun_out/r6_cfg_$name; rm -rf $O; mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python bench.py "$@" --no-cpu-baseline --no-roofline --steps 20 --warmup 5 > $O/log.txt 2>&1
T=$(ls -S $(find $O/p -name "*kernel_trace.csv") | head -1)
python scripts/step_timeline.py $T $O/step_timeline.json > $O/step_timeline.txt 2>&1
cp $(ls -S $(find $O/p -name "*kernel_stats.csv") | head -1) $O/kernel_stats.csv; rm -rf $O/p
python - $O/step_timeline.json <<'PY'
import json, sys, collections
j = json.load(open(sys.argv[1]))
for q, v in j["queues"].items():
    per = collections.OrderedDict()
    for name, us, wgs, thr, vgpr, lds in v["sequence"]:
        e = per.setdefault(name, [0, 0.0, wgs, thr]); e[0] += 1; e[1] += us
    print("queue", q, v["launches"], "launches", v["busy_us"], "us busy")
    for k, e in sorted(per.items(), key=lambda kv: -kv[1][1])[:45]:
        print("  %3d x %-62s %8.1f us  (%.1f each; %d wgs x %d)" % (e[0], k[:62], e[1], e[1] / e[0], e[2], e[3]))
PY
