cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_block_gpu.py tests/test_ops_gpu.py tests/test_axial_layer_gpu.py -x -q -m gpu 2>&1 | tail -6
timeout 200 python scripts/phase_stamps.py 2>&1 | grep -v "^ " | tail -8
for v in "DEFAULT:" "NO_CONV_WAVE:MEDT_CONV_WAVE=0" "NO_WOPOS_WAVE:MEDT_WOPOS_WAVE=0" "NEITHER:MEDT_CONV_WAVE=0 MEDT_WOPOS_WAVE=0" "NO_BLOCK:MEDT_BLOCK_FUSED=0" "DEFAULT2:"; do
  name=${v%%:*}; envs=${v#*:}
  echo -n "$name "; env $envs timeout 200 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step'],4), j['windows_ms'])"
done
