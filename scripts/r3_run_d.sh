cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3d; rm -rf $O; mkdir -p $O
for ls in 16 32 64; do
  MEDT_BWD_LS=$ls python -m pytest tests/test_axial_layer_gpu.py -x -q 2>&1 | tail -3 > $O/tests_ls$ls.log
  MEDT_BWD_LS=$ls python bench.py --roofline-only > $O/roof_ls$ls.json 2>/dev/null
  MEDT_BWD_LS=$ls python bench.py --no-cpu-baseline --no-roofline > $O/step_ls$ls.json 2>/dev/null
  echo "LS=$ls: $(tail -1 $O/tests_ls$ls.log)"
  python -c "import json; j=json.loads(open('$O/roof_ls$ls.json').read()); print(' bwd_core ms', j['roofline']['bwd_core']['launch_ms'])"
  python -c "import json; j=json.loads(open('$O/step_ls$ls.json').read()); print(' step ms', j['ms_per_step'])"
done
