#!/bin/bash
# Round 6: the launch-to-launch spread of the roofline kernel inside ONE process (round-5 VERDICT, weak #5: 133 - 198 us).  Kernel trace of
# `bench.py --roofline-only` with 200 launches per timing -> per-launch duration against start time, per kernel; plus rocm-smi clocks before / after.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6_spread; rm -rf $O; mkdir -p $O
(rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | head -6) > $O/smi_before.txt
# clock sampler: the current sclk level from sysfs every ~2 ms while the bench runs (amdgpu pp_dpm_sclk: the line with '*')
python - "$O/sclk.txt" <<'PY' &
import glob, sys, time
fs = glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")
out = open(sys.argv[1], "w")
if not fs:
    out.write("no pp_dpm_sclk\n"); sys.exit(0)
t_end = time.time() + 40
while time.time() < t_end:
    try:
        cur = [l.strip() for l in open(fs[0]) if "*" in l]
    except Exception as e:
        cur = [repr(e)]
    out.write("%.4f %s\n" % (time.time(), cur[0] if cur else "?"))
    time.sleep(0.002)
PY
SAMPLER=$!
MEDT_ROOF_ITERS=200 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/p -- python bench.py --roofline-only > $O/log.txt 2>&1
(rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | head -6) > $O/smi_after.txt
T=$(ls -S $(find $O/p -name "*kernel_trace.csv") | head -1)
python - "$T" "$O" <<'PY'
import csv, sys, collections, json, statistics
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
by = collections.defaultdict(list)
for r in rows:
    k = r["Kernel_Name"]
    if "attn_fwd4r" in k or "attn_bwd_sweep" in k or "attn_fwd3" in k:
        by[k.split("(")[0][-60:]].append(((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
out = {}
for k, v in by.items():
    d = [x[1] for x in v]
    if len(d) < 20: continue
    # split into bursts (gaps > 5 ms) and report per-burst statistics + the position inside the burst of the slow launches
    bursts, cur = [], [v[0]]
    for a, b in zip(v, v[1:]):
        if b[0] - (a[0] + a[1] / 1e3) > 5.0: bursts.append(cur); cur = []
        cur.append(b)
    bursts.append(cur)
    out[k] = {"launches": len(d), "min": min(d), "median": statistics.median(d), "max": max(d), "stdev": statistics.pstdev(d),
              "bursts": [{"n": len(b), "t_ms": round(b[0][0], 1), "first5": [round(x[1], 1) for x in b[:5]], "median": round(statistics.median(x[1] for x in b), 1),
                          "last5": [round(x[1], 1) for x in b[-5:]], "max": round(max(x[1] for x in b), 1), "argmax": max(range(len(b)), key=lambda i: b[i][1])} for b in bursts]}
    print(k, {kk: (round(vv, 1) if isinstance(vv, float) else vv) for kk, vv in out[k].items() if kk != "bursts"})
    for bb in out[k]["bursts"]: print("   ", bb)
    if "attn_fwd4r" in k:
        out[k]["series_us"] = [round(x[1], 1) for x in v]
        out[k]["start_ms"] = [round(x[0], 2) for x in v]
json.dump(out, open(sys.argv[2] + "/spread.json", "w"), indent=0)
PY
kill $SAMPLER 2>/dev/null; python - $O/sclk.txt <<'PY'
import sys, collections
c = collections.Counter(); n = 0
for l in open(sys.argv[1]):
    p = l.split(None, 1)
    if len(p) == 2: c[p[1].strip()] += 1; n += 1
print("sclk samples:", n, dict(c.most_common(8)))
PY
rm -rf $O/p; cat $O/smi_before.txt $O/smi_after.txt
