#!/bin/bash
# Round 6: kernel trace of the replayed step -> per-queue launch SEQUENCE (scripts/step_timeline.py) and chains (step_chains.py)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r6_trace}
rm -rf $O && mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python bench.py --no-cpu-baseline --no-roofline > $O/bench_prof.log 2>&1
T=$(ls -S $(find $O/bench -name "*kernel_trace.csv") | head -1)      # (the box probes are traced child processes: the largest file is the bench)
python scripts/step_timeline.py $T $O/step_timeline.json > $O/step_timeline.txt 2>&1
python scripts/step_chains.py $T $O/step_chains.json > $O/step_chains.txt 2>&1
S=$(ls -S $(find $O/bench -name "*kernel_stats.csv") | head -1); cp $S $O/kernel_stats.csv
find $O -name "*kernel_trace.csv" -size +20M -delete
tail -3 $O/step_chains.txt
