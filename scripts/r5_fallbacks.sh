#!/bin/bash
# Round 5: the fallback paths behind this round's switches still pass the whole-model and layer parity tests (every switch off at once).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_fallbacks
rm -rf $O && mkdir -p $O
export MEDT_WG_V4=0 MEDT_PREFLIP=0 MEDT_CONV_STEM7=0 MEDT_BWD_WIDE=0 MEDT_TWO_BUCKETS=0 MEDT_BLOCK_BWD=0 MEDT_BLOCK8=0 MEDT_BLOCK_PK=0 MEDT_BF16_RAW32=0
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py tests/test_block_gpu.py tests/test_infer_gpu.py tests/test_dist_gpu.py -m gpu -q 2>&1 | tail -6 | tee $O/tests.txt
timeout 300 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('round-4-equivalent switches:', d['ms_per_step'], d['value'])" | tee -a $O/tests.txt
