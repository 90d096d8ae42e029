cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3l
python -m pytest tests -m gpu -x -q -s 2>&1 | grep -E "product error|passed|failed|Error|^E  " | head -40 > gpurun_out/r3l/tests.log; cat gpurun_out/r3l/tests.log
