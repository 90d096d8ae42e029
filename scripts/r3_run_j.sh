cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_axial_layer_gpu.py -x -q 2>&1 | grep -E "^E  |passed|failed" | head -8
python bench.py --roofline-only 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('bwd_core ms', j['roofline']['bwd_core']['launch_ms'])"
