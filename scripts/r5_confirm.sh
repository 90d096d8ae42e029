#!/bin/bash
# Round 5: confirmation of the final tree ON the GPU box (gpurun): full GPU suite, smoke, the default bench line.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_confirm
rm -rf $O && mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -3 | tee $O/gpu_suite.txt
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 600 python bench.py 2>$O/bench.err | tail -1 > $O/bench_line.json
python -c "import json; j=json.load(open('$O/bench_line.json')); print('step', j['ms_per_step'], j['value'], 'fwd', j['fwd_ms_per_image'], 'roof', j['roofline']['frac'])"
