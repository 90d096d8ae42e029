cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_axial_layer_gpu.py tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -6
for v in "DEFAULT:"; do
  name=${v%%:*}; envs=${v#*:}
  echo -n "$name "; env $envs timeout 200 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step'],4), j['windows_ms'])"
done
for v in "GATED_F32:" "GATED_F32_TWOPASS:MEDT_BWD_SWEEP=0" "GATED_BF16:"; do
  name=${v%%:*}; envs=${v#*:}
  dt=f32; [ "$name" = "GATED_BF16" ] && dt=bf16
  echo -n "$name "; env $envs timeout 200 python bench.py --model gatedaxialunet --batch 8 --dtype $dt --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step'],4), j['windows_ms'])"
done
echo -n "MEDT256 "; timeout 200 python bench.py --model MedT --imgsize 256 --batch 2 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step'],4), j['windows_ms'])"
