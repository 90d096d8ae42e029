#!/bin/bash
# The GPU test files, unchanged, on the CPU lane emulator (tests/lane_emu, tests/emu_device.py): pytest -m gpu --emulate.
# No GPU needed; ~20 minutes on 8 cores (the files run side by side, each emulation is single-threaded).
#   scripts/emu_gpu_suite.sh [outdir]        -> <outdir>/emu_*.log + a summary on stdout   (profiles/r04_emulator_report.txt)
# Not emulated: hipGraph capture (trainer.TrainStep), two-stream scheduling, RCCL, tests that start a GPU subprocess.
set -u
cd "$(dirname "$0")/.."
out=${1:-/tmp/emu_gpu_suite}
mkdir -p "$out"
python - <<'PY'                                     # build libmedt_emu.so once, before the parallel runs
import sys; sys.path[:0] = ["tests", "medical-transformer_amd", "."]
import test_lane_emu as T
print(T.build_emulator())
PY
run() { # name, env, pytest args...
  local name=$1 env=$2; shift 2
  ( env $env timeout 7200 python -m pytest "$@" -m gpu --emulate -q -p no:cacheprovider > "$out/emu_$name.log" 2>&1; echo "$name: $(tail -1 "$out/emu_$name.log")" ) &
}
run ops ""                      tests/test_ops_gpu.py
run block "MEDT_BLOCK_BWD=1"    tests/test_block_gpu.py
run block_v2 "MEDT_BLOCK_BWD=1 MEDT_BLOCK_PK=1" tests/test_block_gpu.py
run block8 "MEDT_BLOCK8=1 MEDT_BLOCK_BWD=1" tests/test_block_gpu.py
run layers ""                   tests/test_axial_layer_gpu.py
run models ""                   tests/test_model_gpu.py -k "test_model_vs_reference_fixture and (axialunet_S64 or S128_N2 or logo)"
run medt_n4_new_kernels "MEDT_BLOCK8=1 MEDT_BLOCK_BWD=1 MEDT_BLOCK_PK=1" tests/test_model_gpu.py -k "test_model_vs_reference_fixture and MedT_S128_N4"
wait
