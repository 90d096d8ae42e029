#!/bin/bash
# Round-4 collection ON the GPU box (gpurun): full GPU test suite + parity report, smoke, the bench lines, rocprofv3 kernel
# statistics of the step and of the roofline legs, the one-switch A/B table of this round's changes, the per-chain summary.
# (The attention kernels' PMC passes are round 3's: those kernels did not change; scripts/collect_profiles.sh re-collects them.)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/profiles_raw
rm -rf $O && mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -E "product error|label map|rel err|worst gradient|trajectory|top-5|passed|failed|FAILED" > $O/parity_report.txt
tail -3 $O/parity_report.txt
timeout 200 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 500 python bench.py 2>$O/bench.err | tail -1 > $O/bench_line.json
python -c "import json; j=json.load(open('$O/bench_line.json')); print('step', j['ms_per_step'], j.get('box_probe'))"
timeout 300 python bench.py --model gatedaxialunet --batch 8 --dtype bf16 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_line_gated_bf16.json
timeout 300 python bench.py --model gatedaxialunet --batch 8 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_line_gated_f32.json
timeout 300 python bench.py --model MedT --imgsize 256 --batch 2 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_line_medt256.json
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python bench.py --no-cpu-baseline --no-roofline > $O/bench_prof.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/roofline -- python bench.py --roofline-only > $O/roofline_prof.log 2>&1
T=$(ls -S $O/bench/*/*_kernel_trace.csv | head -1)
python scripts/step_chains.py $T $O/step_chains.json 12 > $O/step_chains.txt 2>&1
python scripts/step_timeline.py $T $O/step_timeline.json > /dev/null 2>&1
for v in "DEFAULT:" "BLOCK_FUSED_OFF:MEDT_BLOCK_FUSED=0" "BN_DGRAD_FUSED_OFF:MEDT_BN_DGRAD_FUSED=0" "EARLY_FIN_OFF:MEDT_EARLY_FIN=0" \
         "MFMA_WGRAD_IMMEDIATE:MEDT_DEFER_MFMA_WGRAD=0" "TWO_PASS_BWD:MEDT_BWD_SWEEP=0" "IMMEDIATE:MEDT_DEFER=0" "ONE_STREAM:MEDT_TWO_STREAMS=0" \
         "ROUND3_EQUIV:MEDT_BLOCK_FUSED=0 MEDT_BN_DGRAD_FUSED=0 MEDT_EARLY_FIN=0 MEDT_DEFER_MFMA_WGRAD=0" "DEFAULT_AGAIN:"; do
  name=${v%%:*}; envs=${v#*:}
  echo -n "$name " >> $O/ab.txt
  env $envs timeout 200 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 >> $O/ab.txt
done
timeout 100 python scripts/phase_stamps.py 2>&1 | grep -v "^/opt" > $O/phase_stamps.txt
scripts/ubench/group_barrier.bin > $O/group_barrier.json 2>&1
find $O -name "*kernel_trace.csv" -size +20M -delete
cut -c1-150 $O/ab.txt; head -40 $O/step_chains.txt
# (the bench line, the step's kernel statistics and its SQ counters: scripts/r4_final.sh -- the raw counter tables exceed gpurun's
#  64 MiB merge limit and are reduced to per-kernel averages on the box there)
