#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6_call6
rm -rf $O && mkdir -p $O
for m in 3 0; do echo "== MEDT_CONV_THIN=$m" | tee -a $O/dbg.txt; MEDT_CONV_THIN=$m timeout 300 python scripts/r6_dbg_thin.py 2>&1 | grep -v amdgpu.ids | tee -a $O/dbg.txt; done
