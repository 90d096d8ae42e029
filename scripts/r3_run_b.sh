cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3b; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/roof -- python bench.py --roofline-only > $O/roof.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/step -- python bench.py --no-cpu-baseline --no-roofline > $O/step.log 2>&1
find $O -name "*kernel_trace.csv" -size +30M -delete
python -m pytest tests/test_axial_layer_gpu.py -x -q 2>&1 | tail -5 > $O/tests.log
tail -2 $O/roof.log; tail -2 $O/step.log; tail -2 $O/tests.log
