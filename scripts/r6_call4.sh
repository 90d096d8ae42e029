#!/bin/bash
# Round 6, call 4: localise the MedT N=2 train-fixture failure of call 3 (thin kernel: plans with two rows per wave were untested)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6_call4
rm -rf $O && mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -k conv_block 2>&1 | tail -15 | tee $O/ops.txt
for v in "DEFAULT:A=1" "THIN_OFF:MEDT_CONV_THIN=0" "BLOCK8_OFF:MEDT_BLOCK8=0"; do
  name=${v%%:*}; envs=${v#*:}
  echo "== $name" | tee -a $O/model.txt
  env $envs timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -q -s -k "fixture and MedT_S128" 2>&1 | grep -E "product error|passed|failed" | tee -a $O/model.txt
done
