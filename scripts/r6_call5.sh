#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6_call5
rm -rf $O && mkdir -p $O
for v in "FWD_ONLY:MEDT_CONV_THIN=1" "DGRAD_ONLY:MEDT_CONV_THIN=2" "BOTH_TR1:MEDT_THIN_TR=1" "BOTH_TR4:MEDT_THIN_TR=4" "FWD_TR1:MEDT_CONV_THIN=1 MEDT_THIN_TR=1" "DGRAD_TR1:MEDT_CONV_THIN=2 MEDT_THIN_TR=1"; do
  name=${v%%:*}; envs=${v#*:}
  echo "== $name" | tee -a $O/model.txt
  env $envs timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -q -s -k "fixture and MedT_S128_N2_train" 2>&1 | grep -E "product error|passed|failed" | tee -a $O/model.txt
done
