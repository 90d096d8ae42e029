#!/bin/bash
# First GPU call for the one-launch block backward (written in round 4 without GPU minutes left; verified on the CPU lane
# emulator only): parity of the compiled kernel, then the step A/B.  Usage (from the repo root, on the GPU box):
#   bash scripts/r5_block_bwd.sh            -> gpurun_out/r5_block_bwd.txt
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/r5_block_bwd.txt
: > $out
# first, under SHORT timeouts, the new kernels alone (they have never run on a GPU: a hang must not eat the box) -- stop if they fail
for env in "MEDT_BLOCK_BWD=1" "MEDT_BLOCK_BWD=1 MEDT_BLOCK_PK=1" "MEDT_BLOCK8=1 MEDT_BLOCK_BWD=1" "MEDT_BLOCK8=1 MEDT_BLOCK_BWD=1 MEDT_BLOCK_PK=1"; do
  echo "== block kernels alone: $env" >> $out
  if ! env $env timeout 240 python -m pytest tests/test_block_gpu.py -m gpu -x -q 2>&1 | tail -3 >> $out; then echo "STOP: $env failed" >> $out; cat $out; exit 1; fi
  if ! tail -1 $out | grep -q " passed"; then echo "STOP: $env did not pass" >> $out; cat $out; exit 1; fi
done
echo "== parity, MEDT_BLOCK_BWD=1" >> $out
MEDT_BLOCK_BWD=1 timeout 900 python -m pytest tests/test_block_gpu.py tests/test_model_gpu.py tests/test_dist_gpu.py -m gpu -x -q 2>&1 | tail -15 >> $out
echo "== smoke, MEDT_BLOCK_BWD=1" >> $out
MEDT_BLOCK_BWD=1 timeout 600 python __graft_entry__.py --smoke 2>&1 | tail -4 >> $out
echo "== parity, MEDT_BLOCK_BWD=1 MEDT_BLOCK_PK=1 (second-generation instantiations of both block kernels: packed FMAs + transposed wave reductions)" >> $out
MEDT_BLOCK_BWD=1 MEDT_BLOCK_PK=1 timeout 900 python -m pytest tests/test_block_gpu.py tests/test_model_gpu.py -m gpu -x -q 2>&1 | tail -5 >> $out
echo "== parity, MEDT_BLOCK8=1 (8x8-map block forward)" >> $out
MEDT_BLOCK8=1 timeout 900 python -m pytest tests/test_block_gpu.py tests/test_model_gpu.py -m gpu -x -q -k "block8 or test_model_vs_reference_fixture" 2>&1 | tail -5 >> $out
for rep in 1; do
  echo "== bench MEDT_BLOCK8=1 alone (rep $rep)" >> $out
  MEDT_BLOCK8=1 timeout 600 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" >> $out
  echo "== bench everything on: MEDT_BLOCK8=1 MEDT_BLOCK_BWD=1 MEDT_BLOCK_PK=1 (rep $rep)" >> $out
  MEDT_BLOCK8=1 MEDT_BLOCK_BWD=1 MEDT_BLOCK_PK=1 timeout 600 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" >> $out
  for v in "0 0" "0 1" "1 0" "1 1"; do
    set -- $v
    echo "== bench MEDT_BLOCK_BWD=$1 MEDT_BLOCK_PK=$2 (rep $rep)" >> $out
    MEDT_BLOCK_BWD=$1 MEDT_BLOCK_PK=$2 timeout 600 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d.get('launches'))" >> $out
  done
done
echo "== per-kernel timing, all variants" >> $out
timeout 300 python scripts/block_kernels_bench.py 2>&1 | tail -10 >> $out
echo "== phase stamps of the block kernels (libmedt_stamps.so must have been built here: python scripts/phase_stamps.py --build)" >> $out
for pk in 0 1; do echo "-- MEDT_BLOCK_PK=$pk" >> $out; MEDT_BLOCK_BWD=1 MEDT_BLOCK_PK=$pk timeout 300 python scripts/phase_stamps.py 2>&1 | tail -28 >> $out; done
cat $out
