cd "$GRAFT_REPO_ROOT"
MEDT_BWD_LS=32 python -m pytest tests/test_axial_layer_gpu.py -x -q -k "dynamic-16-64 or plain-16-64 or dispatch or reproducible" 2>&1 | tail -2
for nw in 2 4; do MEDT_BWD_NW=$nw MEDT_BWD_LS=32 python bench.py --roofline-only 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read())['roofline']; print('LS32 nw$nw bwd', j['bwd_core']['launch_ms'])"; done
python bench.py --roofline-only 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read())['roofline']; print('LS16 bwd', j['bwd_core']['launch_ms'])"
