#!/bin/bash
# Round 6, first GPU call: the new parity tests (MedT-256 train/evalgrad fixtures, bf16 at the factory state, smoke with the
# factory-state leg), the round's baseline bench line on this box, then the whole GPU suite.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6_call1
rm -rf $O && mkdir -p $O
timeout 300 python -m pytest tests/test_model_gpu.py -m gpu -q -x -s -k "S256 or medt_256 or factory" 2>&1 | tail -40 > $O/new_parity.txt; tail -5 $O/new_parity.txt
timeout 120 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -4 $O/smoke.log
timeout 200 python bench.py 2>$O/bench.err | tail -1 > $O/bench_line.json
python -c "import json; j=json.load(open('$O/bench_line.json')); r=j['roofline']; print('step', j['ms_per_step'], j['value'], 'fwd', j['fwd_ms_per_image'], 'roof', r['frac'], r['valu_frac'], r['launch_ms'], 'bwd', r['bwd_core']['frac'], 'also', [(a['frac']) for a in r.get('also', [])])"
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 | tee $O/gpu_suite.txt
