# round 4: tests named in $AB_TESTS, then bench with one switch flipped per run ($EXTRA_AB = "NAME:ENV=V ENV2=V ..."), then the
# kernel trace of the default configuration + its chain summary (run ON the GPU box through gpurun)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${AB_OUT:-r4_ab}
rm -rf $O && mkdir -p $O
scripts/ubench/clock_probe.bin > $O/clock_probe.json 2>&1
timeout 900 python -m pytest ${AB_TESTS:-tests/test_ops_gpu.py tests/test_model_gpu.py} -m gpu -x -q 2>&1 | tail -15 > $O/tests.txt
IFS=';' read -ra VARS <<< "DEFAULT:;${EXTRA_AB}"
for v in "${VARS[@]}"; do
  [ -z "$v" ] && continue
  name=${v%%:*}; envs=${v#*:}
  echo -n "$name " >> $O/ab.txt
  env $envs timeout 200 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step'],4), j['windows_ms'])" >> $O/ab.txt 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python bench.py --no-cpu-baseline --no-roofline > $O/bench_prof.log 2>&1
T=$(ls -S $O/bench/*/*_kernel_trace.csv | head -1)
python scripts/step_chains.py $T $O/step_chains.json 14 > $O/step_chains.txt 2>&1
python scripts/step_timeline.py $T $O/step_timeline.json > /dev/null 2>&1
find $O -name "*kernel_trace.csv" -size +30M -delete
cat $O/tests.txt; cat $O/ab.txt; head -70 $O/step_chains.txt; cat $O/clock_probe.json | cut -c1-400
