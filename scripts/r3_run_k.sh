cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_dist_gpu.py -x -q 2>&1 | tail -15
grep -o '"ms_per_step": [0-9.]*' gpurun_out/dist_forced_rccl.log
