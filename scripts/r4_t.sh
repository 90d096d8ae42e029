cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_axial_layer_gpu.py -x -q -m gpu -k "conv_block or bit_reproducible or dynamic-128" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "MedT_S128_N4 or deferred or gatedaxialunet_S128_N2" 2>&1 | tail -3
for v in "DEFAULT:" "MFMA_IMMEDIATE:MEDT_DEFER_MFMA_WGRAD=0" "DEFAULT2:"; do
  name=${v%%:*}; envs=${v#*:}
  echo -n "$name "; env $envs timeout 200 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step'],4), j['windows_ms'])"
done
echo -n "GATED_F32 "; timeout 200 python bench.py --model gatedaxialunet --batch 8 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step'],4), j['windows_ms'])"
