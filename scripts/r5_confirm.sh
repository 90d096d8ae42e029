#!/bin/bash
# Round 5: confirmation of the final tree ON the GPU box (gpurun): the default bench line first, smoke, then the full GPU suite
# in whatever is left of the round's GPU minutes (inner timeouts: the box must never be killed from outside).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_confirm
rm -rf $O && mkdir -p $O
timeout 150 python bench.py 2>$O/bench.err | tail -1 > $O/bench_line.json
python -c "import json; j=json.load(open('$O/bench_line.json')); r=j['roofline']; print('step', j['ms_per_step'], j['value'], 'fwd', j['fwd_ms_per_image'], 'roof', r['frac'], r['valu_frac'], r['launch_ms'], 'bwd', r['bwd_core']['frac'], 'also', [(a['frac']) for a in r.get('also', [])])"
timeout 60 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout ${CONFIRM_SUITE_S:-215} python -m pytest tests -m gpu -q -x 2>&1 | tail -3 | tee $O/gpu_suite.txt
