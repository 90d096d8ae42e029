#!/bin/bash
# Round 6, call 3: the thin-channel MFMA convolution kernel + the 8x8 block backward without scratch: parity first, then A/B of the step,
# then the step trace (per-kernel times of the new kernel)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6_call3
rm -rf $O && mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_block_gpu.py -m gpu -q -x 2>&1 | tail -5 | tee $O/ops_block.txt
timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "fixture or factory or medt_256" 2>&1 | tail -5 | tee $O/model.txt
b() { name=$1; shift; echo -n "$name " >> $O/ab.txt; env "$@" timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d.get('fwd_ms_per_image'), d.get('fwd_ms_per_image_bs1'))" >> $O/ab.txt 2>&1; }
b DEFAULT A=1
b THIN_OFF MEDT_CONV_THIN=0
b DEFAULT2 A=1
b THIN_OFF2 MEDT_CONV_THIN=0
cat $O/ab.txt
bash scripts/r6_trace.sh r6_call3/trace
