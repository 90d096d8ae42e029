# quick rocprofv3 kernel statistics of the bench step (run ON the GPU box through gpurun)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_quick
rm -rf $O && mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python bench.py --no-cpu-baseline --no-roofline "$@" > $O/bench_prof.log 2>&1
find $O -name "*kernel_trace.csv" -size +30M -delete
