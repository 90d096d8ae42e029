cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3m
python -m pytest tests -m gpu -x -q -s 2>&1 | grep -E "product error|passed|failed|Error|^E  " | head -40 > gpurun_out/r3m/tests.log; cat gpurun_out/r3m/tests.log
python bench.py --roofline-only 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('bwd_core', j['roofline']['bwd_core'])"
