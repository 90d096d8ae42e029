#!/bin/bash
# Round 5, GPU call 1: the kernels that had never run on an MI355X (one-launch block backward, 8x8 block fwd/bwd, packed-FMA
# instantiations) -- alone under short timeouts, then the step A/B per switch on this one box, then whole-model parity with the
# candidates on.   -> gpurun_out/r5_call1.txt
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/r5_call1.txt
: > $out
for env in "MEDT_BLOCK_BWD=1" "MEDT_BLOCK_BWD=1 MEDT_BLOCK_PK=1" "MEDT_BLOCK8=1" "MEDT_BLOCK8=1 MEDT_BLOCK_BWD=1" "MEDT_BLOCK8=1 MEDT_BLOCK_BWD=1 MEDT_BLOCK_PK=1"; do
  echo "== block kernels alone: $env" >> $out
  env $env timeout 300 python -m pytest tests/test_block_gpu.py -m gpu -q 2>&1 | tail -12 >> $out
done
bench() { env "$@" timeout 300 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d.get('launches'), d.get('fwd_ms_per_image'))" >> $out 2>&1; }
for v in "MEDT_BLOCK_BWD=0" "MEDT_BLOCK_PK=1" "MEDT_BLOCK_BWD=1" "MEDT_BLOCK_BWD=1 MEDT_BLOCK_PK=1" "MEDT_BLOCK8=1" "MEDT_BLOCK8=1 MEDT_BLOCK_PK=1" "MEDT_BLOCK8=1 MEDT_BLOCK_BWD=1" "MEDT_BLOCK8=1 MEDT_BLOCK_BWD=1 MEDT_BLOCK_PK=1" "MEDT_BLOCK_BWD=0"; do
  echo "== bench $v" >> $out
  bench $v
done
echo "== whole-model parity + smoke, everything on" >> $out
MEDT_BLOCK8=1 MEDT_BLOCK_BWD=1 MEDT_BLOCK_PK=1 timeout 900 python -m pytest tests/test_model_gpu.py tests/test_dist_gpu.py -m gpu -x -q 2>&1 | tail -6 >> $out
MEDT_BLOCK8=1 MEDT_BLOCK_BWD=1 MEDT_BLOCK_PK=1 timeout 600 python __graft_entry__.py --smoke 2>&1 | tail -4 >> $out
echo "== whole-model parity, MEDT_BLOCK8=1 MEDT_BLOCK_BWD=1 (first-generation instantiations)" >> $out
MEDT_BLOCK8=1 MEDT_BLOCK_BWD=1 timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -x -q 2>&1 | tail -4 >> $out
echo "== per-kernel timing, all variants" >> $out
timeout 300 python scripts/block_kernels_bench.py 2>&1 | tail -14 >> $out
cat $out
