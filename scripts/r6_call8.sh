#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6_call8
rm -rf $O && mkdir -p $O
echo "== MEDT_CONV_THIN=3 (blocked summation)" | tee -a $O/dbg.txt; timeout 300 python scripts/r6_dbg_thin.py 2>&1 | grep -v amdgpu.ids | tee -a $O/dbg.txt
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -k conv_block 2>&1 | tail -3 | tee $O/ops.txt
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -s -k "fixture or factory or medt_256" 2>&1 | grep -E "product error|passed|failed|rel err" | tee $O/model.txt
b() { name=$1; shift; echo -n "$name " >> $O/ab.txt; env "$@" timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d.get('fwd_ms_per_image'), d.get('fwd_ms_per_image_bs1'))" >> $O/ab.txt 2>&1; }
b DEFAULT A=1
b THIN_OFF MEDT_CONV_THIN=0
b KG1 MEDT_THIN_KG=1
b DEFAULT2 A=1
cat $O/ab.txt
bash scripts/r6_trace.sh r6_call8/trace
