cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r2a
rm -rf $O && mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python bench.py --no-cpu-baseline --no-roofline > $O/bench_prof.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/roofline -- python bench.py --roofline-only > $O/roofline_prof.log 2>&1
MEDT_ROOF_AXIS=h timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/roofline_h -- python bench.py --roofline-only > $O/roofline_h_prof.log 2>&1
find $O -name "*kernel_trace.csv" -size +20M -delete
find $O -name "*_kernel_stats.csv" | head
