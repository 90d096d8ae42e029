cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3a
python bench.py --no-cpu-baseline > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3a/roof -- python bench.py --roofline-only > gpurun_out/r3a/roof.log 2>&1
find gpurun_out/r3a -name "*kernel_trace.csv" -size +30M -delete
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r3a/tests.log
cat gpurun_out/r3a/bench.json; tail -3 gpurun_out/r3a/tests.log
