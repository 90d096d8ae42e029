# A/B of the round-2 late kernels ON the GPU box (gpurun): conv tests, bench with each switch off, kernel statistics
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/ab_quick
rm -rf $O && mkdir -p $O
timeout 900 python -m pytest ${AB_TESTS:-tests/test_ops_gpu.py tests/test_model_gpu.py} -m gpu -x -q 2>&1 | tail -15 > $O/tests.txt
for v in "DEFAULT:" $EXTRA_AB; do
  name=${v%%:*}; envs=${v#*:}
  echo -n "$name " >> $O/ab.txt
  env $envs timeout 200 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 >> $O/ab.txt
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python bench.py --no-cpu-baseline --no-roofline > $O/bench_prof.log 2>&1
find $O -name "*kernel_trace.csv" -size +30M -delete
cat $O/tests.txt; cat $O/ab.txt
if [ -n "$ROOF" ]; then timeout 300 python bench.py --roofline-only 2>/dev/null | tail -1 > $O/roofline.json; cat $O/roofline.json | cut -c1-1500; fi
