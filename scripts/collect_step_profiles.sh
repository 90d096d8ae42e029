#!/bin/bash
# The step-level half of scripts/collect_profiles.sh (bench lines, kernel statistics of the step, the one-switch A/B table)
# -- re-run on its own when the first collection landed on a slow box (bench.py's box_probe tells them apart).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/profiles_raw
mkdir -p $O
rm -rf $O/bench $O/ab.txt
timeout 500 python bench.py 2>$O/bench.err | tail -1 > $O/bench_line.json
python -c "import json; j=json.load(open('$O/bench_line.json')); print('step', j['ms_per_step'], j.get('box_probe'))"
timeout 300 python bench.py --model gatedaxialunet --batch 8 --dtype bf16 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_line_gated_bf16.json
timeout 300 python bench.py --model gatedaxialunet --batch 8 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_line_gated_f32.json
timeout 300 python bench.py --model MedT --imgsize 256 --batch 2 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_line_medt256.json
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python bench.py --no-cpu-baseline --no-roofline > $O/bench_prof.log 2>&1
for v in "DEFAULT:" "BN_FIN_APPLY_OFF:MEDT_BN_FIN_APPLY=0" "BN_CHAN_OFF:MEDT_BN_CHAN_MAX=0" "WGRAD_R2_CHUNKS:MEDT_WG_CHUNKS=32 MEDT_WG_QMAX=512" \
         "CONV_WS_OFF:MEDT_FWD_WS=0 MEDT_DGRAD_WS_POS3=4096" "TWO_PASS_BWD:MEDT_BWD_SWEEP=0" "UP2X_SCALAR:MEDT_UP2X_VEC=0" "WGRAD_TILE64:MEDT_WG_TILE=64" "VALU_WGRAD:MEDT_WGRAD_VALU=1" \
         "IMMEDIATE:MEDT_DEFER=0" "ONE_STREAM:MEDT_TWO_STREAMS=0" "NO_SPLIT_FLUSH:MEDT_SPLIT_FLUSH=0" "DEFAULT_AGAIN:"; do
  name=${v%%:*}; envs=${v#*:}
  echo -n "$name " >> $O/ab.txt
  env $envs timeout 200 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 >> $O/ab.txt
done
timeout 120 python scripts/graph_host_cost.py 2>&1 | grep -v Warning | grep -E "replays|GPU-event|single|Model name|clk|Power" > $O/graph_host_cost.txt
find $O -name "*kernel_trace.csv" -size +20M -delete
