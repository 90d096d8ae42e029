#!/bin/bash
# Round 6, call 18: channel_sum (32-bit indices, adaptive splits) + reduce_rows (16 chains): ops parity, bench, trace
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r6_call18}
rm -rf $O && mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x 2>&1 | tail -4 | tee $O/ops.txt
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "graphed or deferred or factory_state_train_parity or trajectory or fixture" 2>&1 | tail -4 | tee $O/model.txt
b() { name=$1; shift; echo -n "$name " >> $O/ab.txt; env "$@" timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d.get('fwd_ms_per_image'), d.get('fwd_ms_per_image_bs1'))" >> $O/ab.txt 2>&1; }
b DEFAULT A=1
b DEFAULT2 A=1
cat $O/ab.txt
bash scripts/r6_trace.sh $(basename $O)/trace
python - <<PY
import json
b=json.load(open('$O/trace/step_timeline.json'))
for q,v in b['queues'].items():
    for s in v['sequence']:
        if 'channel_sum' in s[0] or 'reduce_rows' in s[0] or 'wgrad' in s[0]: print(q, s)
PY
