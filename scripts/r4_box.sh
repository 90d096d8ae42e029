# round 4: what separates the slow boxes?  (run ON the GPU box through gpurun)
#   1. both box probes + the group-barrier microbenchmark
#   2. rocm-smi clocks / power / perf level, idle and while the light-load chain runs
#   3. a short bench at the box's own settings, then with the performance level forced high (if the box lets us), then restored
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_box
rm -rf $O && mkdir -p $O
U=scripts/ubench
$U/load_latency.bin > $O/load_latency.json 2>&1
$U/clock_probe.bin > $O/clock_probe_auto.json 2>&1
$U/group_barrier.bin > $O/group_barrier.json 2>&1
rocm-smi --showperflevel --showclocks --showpower --showtemp > $O/smi_idle.txt 2>&1
( for i in 1 2 3 4 5 6; do $U/clock_probe.bin > /dev/null 2>&1; done ) &
sleep 1; rocm-smi --showclocks --showpower > $O/smi_light_load.txt 2>&1; wait
timeout 200 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_auto.json
( timeout 100 python bench.py --no-cpu-baseline --no-roofline --steps 200 >/dev/null 2>&1 ) &
sleep 25; rocm-smi --showclocks --showpower > $O/smi_bench_load.txt 2>&1; wait
rocm-smi --setperflevel high > $O/setperf.txt 2>&1
rocm-smi --showperflevel >> $O/setperf.txt 2>&1
$U/clock_probe.bin > $O/clock_probe_high.json 2>&1
timeout 200 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_high.json
rocm-smi --setperfdeterminism 2400 >> $O/setperf.txt 2>&1
$U/clock_probe.bin > $O/clock_probe_determinism.json 2>&1
timeout 200 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_determinism.json
rocm-smi --resetperfdeterminism >> $O/setperf.txt 2>&1
rocm-smi --setperflevel auto >> $O/setperf.txt 2>&1
for f in load_latency clock_probe_auto group_barrier clock_probe_high clock_probe_determinism; do echo "== $f"; cat $O/$f.json; done
for f in bench_auto bench_high bench_determinism; do echo "== $f"; python -c "import json,sys; j=json.load(open('$O/$f.json')); print(j['ms_per_step'], j['windows_ms'])"; done
cat $O/setperf.txt | tail -20
grep -E "sclk|mclk|fclk|Power|perf" -i $O/smi_idle.txt $O/smi_light_load.txt $O/smi_bench_load.txt | head -40
