#!/bin/bash
# Round 6, call 11: the last flush's MFMA weight gradients forked onto the idle branch stream (medt_queue_flush2)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6_call11
rm -rf $O && mkdir -p $O
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "graphed or deferred or failed_step or flat_adam or factory_state_train_parity or trajectory" 2>&1 | tail -4 | tee $O/model.txt
b() { name=$1; shift; echo -n "$name " >> $O/ab.txt; env "$@" timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d.get('fwd_ms_per_image'), d.get('fwd_ms_per_image_bs1'))" >> $O/ab.txt 2>&1; }
b DEFAULT A=1
b TAIL_FORK_OFF MEDT_TAIL_FORK=0
b DEFAULT2 A=1
b TAIL_FORK_OFF2 MEDT_TAIL_FORK=0
cat $O/ab.txt
