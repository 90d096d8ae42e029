cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3f
./scripts/ubench/valu_rate.bin > gpurun_out/r3f/valu_rate.txt 2>&1; cat gpurun_out/r3f/valu_rate.txt
python -m pytest tests/test_axial_layer_gpu.py -x -q 2>&1 | grep -E "^E|assert|Error|bad" | head -20
