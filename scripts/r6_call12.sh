#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6_call12
rm -rf $O && mkdir -p $O
b() { name=$1; shift; echo -n "$name " >> $O/ab.txt; env "$@" timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d.get('fwd_ms_per_image'), d.get('fwd_ms_per_image_bs1'))" >> $O/ab.txt 2>&1; }
b DEFAULT A=1
b THIN_TR2 MEDT_THIN_TR=2
b THIN_TR1 MEDT_THIN_TR=1
b DEFAULT2 A=1
cat $O/ab.txt
bash scripts/r6_trace.sh r6_call12/trace
MEDT_THIN_TR=1 bash scripts/r6_trace.sh r6_call12/trace_tr1
