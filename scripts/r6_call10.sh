#!/bin/bash
# Round 6, call 9: the stride-2 block forward on the MI355X: parity first, then the step A/B and the trace
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6_call10
rm -rf $O && mkdir -p $O
timeout 600 python -m pytest tests/test_block_gpu.py -m gpu -q 2>&1 | tail -5 | tee $O/block.txt
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -s -k "(fixture and MedT) or factory_state_train_parity" 2>&1 | grep -E "product error|passed|failed|rel err" | tee $O/model.txt
b() { name=$1; shift; echo -n "$name " >> $O/ab.txt; env "$@" timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d.get('fwd_ms_per_image'), d.get('fwd_ms_per_image_bs1'))" >> $O/ab.txt 2>&1; }
b DEFAULT A=1
b S2_OFF MEDT_BLOCK_S2=0
b DEFAULT2 A=1
b MFMA_OFF MEDT_BLOCK_MFMA=0
cat $O/ab.txt
bash scripts/r6_trace.sh r6_call10/trace
