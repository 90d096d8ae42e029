#!/bin/bash
# Round 5, GPU call 5: the fixed 16-byte weight-gradient body on the hardware, the full GPU suite, A/B of its chunking and of the
# wide-lane sweeps (MEDT_BWD_WIDE), MedT-256 / gated with wide = 2.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_call5
rm -rf $O && mkdir -p $O
timeout 200 python scripts/r5_dbg_v4.py 2>&1 | grep -v amdgpu.ids > $O/dbg.txt; cat $O/dbg.txt
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -E "product error|label map|rel err|worst gradient|trajectory|top-5|passed|failed|FAILED|Error|eval forward|factory state|assert" > $O/parity_report.txt
tail -12 $O/parity_report.txt
b() { name=$1; shift; echo -n "$name " >> $O/ab.txt; env "$@" timeout 300 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" >> $O/ab.txt 2>&1; }
b DEFAULT MEDT_X=0
b BWD_WIDE_OFF MEDT_BWD_WIDE=0
b WG4_32_512 MEDT_WG4_CHUNKS=32 MEDT_WG4_QMAX=512
b WG4_64_256 MEDT_WG4_CHUNKS=64 MEDT_WG4_QMAX=256
b WG4_32_256 MEDT_WG4_CHUNKS=32 MEDT_WG4_QMAX=256
b WG4_64_512 MEDT_WG4_CHUNKS=64 MEDT_WG4_QMAX=512
b WG_V4_OFF MEDT_WG_V4=0
b DEFAULT_AGAIN MEDT_X=0
cat $O/ab.txt
for cfg in "medt256 MEDT_X=0 --model MedT --imgsize 256 --batch 2" "medt256_wide2 MEDT_BWD_WIDE=2 --model MedT --imgsize 256 --batch 2" "gated_f32 MEDT_X=0 --model gatedaxialunet --batch 8" "gated_f32_wide2 MEDT_BWD_WIDE=2 --model gatedaxialunet --batch 8" "gated_bf16 MEDT_X=0 --model gatedaxialunet --batch 8 --dtype bf16"; do
  set -- $cfg; name=$1; e=$2; shift; shift
  env $e timeout 300 python bench.py "$@" --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_line_$name.json
  python -c "import json; j=json.load(open('$O/bench_line_$name.json')); print('$name', j['ms_per_step'], j['value'], j.get('fwd_ms_per_image'))"
done
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python bench.py --no-cpu-baseline --no-roofline > $O/bench_prof.log 2>&1
T=$(ls -S $O/bench/*/*_kernel_trace.csv | head -1)
python scripts/step_chains.py $T $O/step_chains.json 12 > $O/step_chains.txt 2>&1
cp $(ls -S $O/bench/*/*_kernel_stats.csv | head -1) $O/bench_kernel_stats.csv; rm -rf $O/bench
grep -E "wgrad|reduce_rows|sweep" $O/bench_kernel_stats.csv | cut -c1-160
du -sh $O
