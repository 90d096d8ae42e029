#!/bin/bash
# Round 5, GPU call 2: full GPU suite (new: InferStep, factory-state parity, gp=4 / L=128 sweep cases, un-skipped block kernels),
# bench lines of all BASELINE configurations, kernel statistics, and the FETCH_SIZE / WRITE_SIZE passes behind roofline_traffic.json.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_call2
rm -rf $O && mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -s -x 2>&1 | grep -E "product error|label map|rel err|worst gradient|trajectory|top-5|passed|failed|FAILED|Error|eval forward|factory state|assert" > $O/parity_report.txt
tail -6 $O/parity_report.txt
timeout 500 python bench.py 2>$O/bench.err | tail -1 > $O/bench_line.json
python -c "import json; j=json.load(open('$O/bench_line.json')); print('step', j['ms_per_step'], 'fwd/img', j.get('fwd_ms_per_image'), j.get('fwd_ms_per_image_bs1'), j.get('fwd_ms_per_image_eager')); r=j['roofline']; print('roof', r['frac'], r['bwd_core']['frac'], r['also'])"
timeout 300 python bench.py --model MedT --imgsize 256 --batch 2 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_line_medt256.json
timeout 300 python bench.py --model gatedaxialunet --batch 8 --dtype bf16 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_line_gated_bf16.json
timeout 300 python bench.py --model gatedaxialunet --batch 8 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_line_gated_f32.json
for f in medt256 gated_bf16 gated_f32; do python -c "import json; j=json.load(open('$O/bench_line_$f.json')); print('$f', j['ms_per_step'], j['value'], j.get('fwd_ms_per_image'))"; done
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python bench.py --no-cpu-baseline --no-roofline > $O/bench_prof.log 2>&1
T=$(ls -S $O/bench/*/*_kernel_trace.csv | head -1)
python scripts/step_chains.py $T $O/step_chains.json 12 > $O/step_chains.txt 2>&1
cp $(ls -S $O/bench/*/*_kernel_stats.csv | head -1) $O/bench_kernel_stats.csv; rm -rf $O/bench
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/m256 -- python bench.py --model MedT --imgsize 256 --batch 2 --no-cpu-baseline --no-roofline > $O/m256_prof.log 2>&1
cp $(ls -S $O/m256/*/*_kernel_stats.csv | head -1) $O/m256_kernel_stats.csv; rm -rf $O/m256
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/roofline -- python bench.py --roofline-only > $O/roofline_prof.log 2>&1
cp $(ls -S $O/roofline/*/*_kernel_stats.csv | head -1) $O/roofline_kernel_stats.csv; rm -rf $O/roofline
grep '^{"roofline' $O/roofline_prof.log | tail -1 > $O/roofline_only.json
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python bench.py --roofline-only > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python bench.py --roofline-only > $O/pmc_write.log 2>&1
python scripts/r5_traffic.py $O/pmc_fetch $O/pmc_write $O/roofline_only.json "$(cat .commit_stamp 2>/dev/null)" > $O/roofline_traffic.json 2>$O/traffic.err; tail -2 $O/traffic.err
rm -rf $O/pmc_fetch $O/pmc_write
head -12 $O/step_chains.txt | cut -c1-120; grep -E "sweep|attn_bwd_kernel|attn_bwd_stats" $O/m256_kernel_stats.csv | cut -c1-200 | head
du -sh $O
