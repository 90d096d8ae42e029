#!/bin/bash
# Round 5, GPU call 4 (debugging): the 16-byte weight-gradient body fails on the MI355X for tiles with one 16-column block
# (it passes on the emulator): error patterns of the default build and of two debugging variants; dist tests with full output.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_call4
rm -rf $O && mkdir -p $O
for v in default b32 nops; do
  echo "=== variant $v" >> $O/dbg.txt
  if [ $v = default ]; then timeout 200 python scripts/r5_dbg_v4.py >> $O/dbg.txt 2>&1
  else MEDT_LIB_OVERRIDE=$PWD/medical-transformer_amd/libmedt_dbg_$v.so timeout 200 python scripts/r5_dbg_v4.py >> $O/dbg.txt 2>&1; fi
done
grep -v "amdgpu.ids" $O/dbg.txt
timeout 600 python -m pytest tests/test_dist_gpu.py tests/test_infer_gpu.py -m gpu -q -x -s 2>&1 | tail -40 > $O/dist.txt; tail -25 $O/dist.txt
