# round-4 baseline: GPU tests, bench line, step kernel statistics (run ON the GPU box through gpurun)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_base
rm -rf $O && mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/tests.txt
timeout 400 python bench.py 2>$O/bench.err | tail -1 > $O/bench_line.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python bench.py --no-cpu-baseline --no-roofline > $O/bench_prof.log 2>&1
find $O -name "*kernel_trace.csv" -size +30M -delete
cat $O/tests.txt; cut -c1-600 $O/bench_line.json
