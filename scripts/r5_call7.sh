#!/bin/bash
# Round 5, GPU call 7: the 3x3 instance of the 16-byte weight-gradient body on the hardware (parity first), its A/B on all configs,
# a scan of the forward kernel's grid knobs on the roofline shape, then the full suite.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_call7
rm -rf $O && mkdir -p $O
timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "recorded" 2>&1 | tail -5 | tee $O/recorded.txt
b() { name=$1; shift; echo -n "$name " >> $O/ab.txt; env "$@" timeout 300 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" >> $O/ab.txt 2>&1; }
b DEFAULT MEDT_X=0
b K3_OFF MEDT_WG_V4_K3=0
b DEFAULT_AGAIN MEDT_X=0
b K3_OFF_AGAIN MEDT_WG_V4_K3=0
cat $O/ab.txt
for cfg in "medt256 MEDT_X=0 --model MedT --imgsize 256 --batch 2" "medt256_k3off MEDT_WG_V4_K3=0 --model MedT --imgsize 256 --batch 2" "gated_f32 MEDT_X=0 --model gatedaxialunet --batch 8" "gated_f32_k3off MEDT_WG_V4_K3=0 --model gatedaxialunet --batch 8"; do
  set -- $cfg; name=$1; e=$2; shift; shift
  env $e timeout 300 python bench.py "$@" --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_line_$name.json
  python -c "import json; j=json.load(open('$O/bench_line_$name.json')); print('$name', j['ms_per_step'], j['value'], j.get('fwd_ms_per_image'))"
done
r() { name=$1; shift; echo -n "$name " >> $O/roof_scan.txt; env "$@" timeout 120 python bench.py --roofline-only 2>/dev/null | grep '^{"roofline' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(r['launch_ms']*1e3,1), round(r['frac'],4), 'bwd', round(r['bwd_core']['launch_ms']*1e3,1), round(r['bwd_core']['frac'],4), '| C32L128', round(d['also']['launch_ms']*1e3,1), round(d['also']['frac'],4))" >> $O/roof_scan.txt 2>&1; }
r DEFAULT MEDT_X=0
r CAP1024 MEDT_CAP=1024
r CAP1536 MEDT_CAP=1536
r CAP3072 MEDT_CAP=3072
r CAP4096 MEDT_CAP=4096
r NT1 MEDT_NT=1
r NT2 MEDT_NT=2
cat $O/roof_scan.txt
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python bench.py --no-cpu-baseline --no-roofline > $O/bench_prof.log 2>&1
T=$(ls -S $O/bench/*/*_kernel_trace.csv | head -1)
python scripts/step_chains.py $T $O/step_chains.json 12 > $O/step_chains.txt 2>&1
cp $(ls -S $O/bench/*/*_kernel_stats.csv | head -1) $O/bench_kernel_stats.csv; rm -rf $O/bench
grep -E "wgrad_mfma_grouped|reduce_rows_grouped" $O/bench_kernel_stats.csv | cut -c1-150
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -E "product error|label map|rel err|worst gradient|trajectory|top-5|passed|failed|FAILED|Error|eval forward|factory state|assert" > $O/parity_report.txt
tail -4 $O/parity_report.txt
