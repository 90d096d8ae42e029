cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "deferred or failed_step or graph or MedT_S128_N4 or trajectory" 2>&1 | tail -3
for v in "DEFAULT:" "MFMA_IMM:MEDT_DEFER_MFMA_WGRAD=0" "DEFAULT2:"; do
  name=${v%%:*}; envs=${v#*:}
  echo -n "$name "; env $envs timeout 200 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step'],4), round(j['fwd_ms_per_image'],3), j['windows_ms'])"
done
