#!/bin/bash
# Round 5, GPU call 3: full GPU suite again (call 2 stopped at InferStep's train-mode capture), the A/B of this round's flush-tail
# work (16-byte K=1 weight-gradient body, merged tail launch), bf16 with the fused finalize+apply, kernel statistics, HBM traffic.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_call3
rm -rf $O && mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -E "product error|label map|rel err|worst gradient|trajectory|top-5|passed|failed|FAILED|Error|eval forward|factory state|assert" > $O/parity_report.txt
tail -8 $O/parity_report.txt
b() { name=$1; shift; echo -n "$name " >> $O/ab.txt; env "$@" timeout 300 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" >> $O/ab.txt 2>&1; }
b DEFAULT MEDT_X=0
b WG_V4_OFF MEDT_WG_V4=0
b WG_V4_OFF_TAIL_MERGED MEDT_WG_V4=0 MEDT_WGRAD_TAIL=1
b WG4_CHUNKS8 MEDT_WG4_CHUNKS=8
b WG4_CHUNKS32_QMAX512 MEDT_WG4_CHUNKS=32 MEDT_WG4_QMAX=512
b DEFAULT_AGAIN MEDT_X=0
cat $O/ab.txt
for cfg in "gated_bf16 --model gatedaxialunet --batch 8 --dtype bf16" "gated_f32 --model gatedaxialunet --batch 8" "gated_bf16_again --model gatedaxialunet --batch 8 --dtype bf16"; do
  set -- $cfg; name=$1; shift
  timeout 300 python bench.py "$@" --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_line_$name.json
  python -c "import json; j=json.load(open('$O/bench_line_$name.json')); print('$name', j['ms_per_step'], j['value'], j.get('fwd_ms_per_image'))"
done
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python bench.py --no-cpu-baseline --no-roofline > $O/bench_prof.log 2>&1
T=$(ls -S $O/bench/*/*_kernel_trace.csv | head -1)
python scripts/step_chains.py $T $O/step_chains.json 12 > $O/step_chains.txt 2>&1
cp $(ls -S $O/bench/*/*_kernel_stats.csv | head -1) $O/bench_kernel_stats.csv; rm -rf $O/bench
grep -E "wgrad|reduce_rows" $O/bench_kernel_stats.csv | cut -c1-160
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/roofline -- python bench.py --roofline-only > $O/roofline_prof.log 2>&1
grep '^{"roofline' $O/roofline_prof.log | tail -1 > $O/roofline_only.json; rm -rf $O/roofline
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python bench.py --roofline-only > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python bench.py --roofline-only > $O/pmc_write.log 2>&1
python scripts/r5_traffic.py $O/pmc_fetch $O/pmc_write $O/roofline_only.json "$(cat .commit_stamp 2>/dev/null)" > $O/roofline_traffic.json 2>$O/traffic.err; tail -2 $O/traffic.err
rm -rf $O/pmc_fetch $O/pmc_write
python -c "import json; j=json.load(open('$O/roofline_traffic.json')); print({k: j[k] for k in j if 'over' in k or 'bytes' in k}); print(j['C32_L128'])" 2>&1 | cut -c1-600
du -sh $O
