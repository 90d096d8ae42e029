cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_dist_gpu.py -x -q -m gpu 2>&1 | tail -6
for v in "DEFAULT:" "NO_SEAL:MEDT_SEAL=0" "NO_DEFER_MFMA:MEDT_DEFER_MFMA_WGRAD=0" "NEITHER:MEDT_SEAL=0 MEDT_DEFER_MFMA_WGRAD=0" "SKIP_SWEEP:MEDT_SKIP=sweep" "SKIP_WOPOS_BWD:MEDT_SKIP=wopos_bwd" "DEFAULT2:"; do
  name=${v%%:*}; envs=${v#*:}
  echo -n "$name "; env $envs timeout 200 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step'],4), j['windows_ms'])"
done
