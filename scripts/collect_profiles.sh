#!/bin/bash
# Runs ON the GPU box (gpurun): the bench lines, rocprofv3 kernel statistics of the same commands, the PMC passes behind
# profiles/roofline_traffic.json and profiles/rNN_attn_pmc.json, the grouped weight-gradient MFMA/VALU A/B, the host
# thread sweep of the CPU baseline and the parity report.  Outputs land in gpurun_out/profiles_raw/.
# (counter passes: --pmc with --kernel-trace only, one counter group per run -- MI355X_MICROARCH.md)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/profiles_raw
rm -rf $O && mkdir -p $O
timeout 500 python bench.py 2>$O/bench.err | tail -1 > $O/bench_line.json
timeout 300 python bench.py --model gatedaxialunet --batch 8 --dtype bf16 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_line_gated_bf16.json
timeout 300 python bench.py --model gatedaxialunet --batch 8 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_line_gated_f32.json
timeout 300 python bench.py --model MedT --imgsize 256 --batch 2 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_line_medt256.json
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python bench.py --no-cpu-baseline --no-roofline > $O/bench_prof.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/roofline -- python bench.py --roofline-only > $O/roofline_prof.log 2>&1
MEDT_ROOF_AXIS=h timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/roofline_h -- python bench.py --roofline-only > $O/roofline_h_prof.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/roofline_bf16 -- python bench.py --roofline-only --dtype bf16 > $O/roofline_bf16_prof.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python bench.py --roofline-only > $O/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python bench.py --roofline-only > $O/pmc_write.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq1 -- python bench.py --roofline-only > $O/pmc_sq1.log 2>&1
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq2 -- python bench.py --roofline-only > $O/pmc_sq2.log 2>&1
# one switch flipped per run (the round's changes and the older structural switches)
for v in "DEFAULT:" "BN_FIN_APPLY_OFF:MEDT_BN_FIN_APPLY=0" "BN_CHAN_OFF:MEDT_BN_CHAN_MAX=0" "WGRAD_R2_CHUNKS:MEDT_WG_CHUNKS=32 MEDT_WG_QMAX=512" \
         "CONV_WS_OFF:MEDT_FWD_WS=0 MEDT_DGRAD_WS_POS3=4096" "TWO_PASS_BWD:MEDT_BWD_SWEEP=0" "UP2X_SCALAR:MEDT_UP2X_VEC=0" "WGRAD_TILE64:MEDT_WG_TILE=64" "VALU_WGRAD:MEDT_WGRAD_VALU=1" \
         "IMMEDIATE:MEDT_DEFER=0" "ONE_STREAM:MEDT_TWO_STREAMS=0" "NO_SPLIT_FLUSH:MEDT_SPLIT_FLUSH=0" "DEFAULT_AGAIN:"; do
  name=${v%%:*}; envs=${v#*:}
  echo -n "$name " >> $O/ab.txt
  env $envs timeout 200 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 >> $O/ab.txt
done
timeout 120 python scripts/graph_host_cost.py 2>&1 | grep -v Warning | tail -12 > $O/graph_host_cost.txt
(cd scripts/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate.bin 2>/dev/null && timeout 60 ./valu_rate.bin) > $O/valu_rate.txt 2>&1
# parity report (product error / reference fp32 noise per gradient tensor, excluded-pixel counts, bf16 errors)
timeout 1200 python -m pytest tests -m gpu -q -s 2>&1 | grep -E "product error|label map|rel err|worst gradient|trajectory|top-5|passed|failed" > $O/parity_report.txt
timeout 120 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
cp gpurun_out/dist_forced_rccl.log $O/ 2>/dev/null
find $O -name "*kernel_trace.csv" -path "*pmc*" -delete
find $O -name "*kernel_trace.csv" -size +20M -delete
ls -R $O | head -60
