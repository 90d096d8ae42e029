#!/bin/bash
# Runs ON the GPU box (gpurun): the bench line, rocprofv3 kernel statistics of the same commands, and the PMC passes
# behind profiles/roofline_traffic.json and profiles/r01_attn_fwd_pmc.json.  Outputs land in gpurun_out/profiles_raw/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/profiles_raw
rm -rf $O && mkdir -p $O
timeout 400 python bench.py 2>$O/bench.err | tail -1 > $O/bench_line.json
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python bench.py --no-cpu-baseline > $O/bench_prof.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/roofline -- python bench.py --roofline-only > $O/roofline_prof.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python bench.py --roofline-only > $O/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python bench.py --roofline-only > $O/pmc_write.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq1 -- python bench.py --roofline-only > $O/pmc_sq1.log 2>&1
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq2 -- python bench.py --roofline-only > $O/pmc_sq2.log 2>&1
find $O -name "*kernel_trace.csv" -path "*pmc*" -delete
find $O -name "*kernel_trace.csv" -size +20M -delete
ls -R $O | head -40
