#!/bin/bash
# Round 6, call 13: backward flushes on flush streams (defer.ASYNC): parity subset + A/B over the trigger threshold + trace
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6_call13
rm -rf $O && mkdir -p $O
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "graphed or deferred or failed_step or flat_adam or factory_state_train_parity or trajectory" 2>&1 | tail -4 | tee $O/model.txt
b() { name=$1; shift; echo -n "$name " >> $O/ab.txt; env "$@" timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d.get('fwd_ms_per_image'), d.get('fwd_ms_per_image_bs1'))" >> $O/ab.txt 2>&1; }
b ASYNC_OFF MEDT_ASYNC_FLUSH=0
b ASYNC_16 A=1
b ASYNC_8 MEDT_ASYNC_MIN=8
b ASYNC_32 MEDT_ASYNC_MIN=32
b ASYNC_64 MEDT_ASYNC_MIN=64
b ASYNC_1000 MEDT_ASYNC_MIN=1000
b ASYNC_OFF2 MEDT_ASYNC_FLUSH=0
b ASYNC_16b A=1
cat $O/ab.txt
bash scripts/r6_trace.sh r6_call13/trace
