cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_axial_layer_gpu.py -x -q -k "dispatch or rows or bound" 2>&1 | tail -2
for i in 1 2; do python bench.py --roofline-only 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read())['roofline']; print('fwd ms', j['launch_ms'], 'frac', j['frac'], 'bwd', j['bwd_core']['launch_ms'])"; done
