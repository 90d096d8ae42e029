#!/bin/bash
# Round 5, GPU call 11: half-width rows16 workgroups: parity, A/B.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_call13
rm -rf $O && mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -q -x 2>&1 | tail -4 | tee $O/tests.txt
b() { name=$1; shift; echo -n "$name " >> $O/ab.txt; env "$@" timeout 300 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d.get('fwd_ms_per_image'))" >> $O/ab.txt 2>&1; }
b DEFAULT MEDT_X=0
b R16W_OT64 MEDT_R16W_OT=64
b DEFAULT_AGAIN MEDT_X=0
b R16W_OT64_AGAIN MEDT_R16W_OT=64
cat $O/ab.txt
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python bench.py --no-cpu-baseline --no-roofline > $O/bench_prof.log 2>&1
cp $(ls -S $O/bench/*/*_kernel_stats.csv | head -1) $O/bench_kernel_stats.csv; rm -rf $O/bench
grep -E "rows16|stem7" $O/bench_kernel_stats.csv | cut -c1-170
