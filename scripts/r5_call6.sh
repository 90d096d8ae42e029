#!/bin/bash
# Round 5, GPU call 6: suite + A/B after the bf16 raw32 path, the merged dedicated weight-gradient launch, gp=4 wide sweeps.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_call6
rm -rf $O && mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -E "product error|label map|rel err|worst gradient|trajectory|top-5|passed|failed|FAILED|Error|eval forward|factory state|assert" > $O/parity_report.txt
tail -6 $O/parity_report.txt
b() { name=$1; shift; echo -n "$name " >> $O/ab.txt; env "$@" timeout 300 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" >> $O/ab.txt 2>&1; }
b DEFAULT MEDT_X=0
b TAIL_OFF MEDT_WGRAD_TAIL=0
b SLABS_BIG_256 MEDT_WG_SLABS_BIG=256
b SLABS_BIG_128 MEDT_WG_SLABS_BIG=128
b DEFAULT_AGAIN MEDT_X=0
cat $O/ab.txt
for cfg in "gated_f32 MEDT_X=0 --model gatedaxialunet --batch 8" "gated_bf16 MEDT_X=0 --model gatedaxialunet --batch 8 --dtype bf16" "gated_bf16_raw32_off MEDT_BF16_RAW32=0 --model gatedaxialunet --batch 8 --dtype bf16" "gated_f32_again MEDT_X=0 --model gatedaxialunet --batch 8" "medt256 MEDT_X=0 --model MedT --imgsize 256 --batch 2"; do
  set -- $cfg; name=$1; e=$2; shift; shift
  env $e timeout 300 python bench.py "$@" --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_line_$name.json
  python -c "import json; j=json.load(open('$O/bench_line_$name.json')); print('$name', j['ms_per_step'], j['value'], j.get('fwd_ms_per_image'))"
done
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python bench.py --no-cpu-baseline --no-roofline > $O/bench_prof.log 2>&1
T=$(ls -S $O/bench/*/*_kernel_trace.csv | head -1)
python scripts/step_chains.py $T $O/step_chains.json 12 > $O/step_chains.txt 2>&1
cp $(ls -S $O/bench/*/*_kernel_stats.csv | head -1) $O/bench_kernel_stats.csv; rm -rf $O/bench
grep -E "wgrad|reduce_rows" $O/bench_kernel_stats.csv | cut -c1-150 | head -8
du -sh $O
