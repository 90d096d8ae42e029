#!/bin/bash
# Timing experiments on the single-sweep backward (NOT product builds): build libmedt_hip.so variants that leave one piece
# of the row loop out (-DMEDT_ABL=n), then time the roofline leg with each through MEDT_LIB_OVERRIDE on the GPU box.
#   here:        bash scripts/r3_ablate.sh build
#   on the box:  bash scripts/r3_ablate.sh run
cd "$(dirname "$0")/.."
C=medical-transformer_amd/csrc
if [ "$1" = build ]; then
  mkdir -p $C/build/abl
  for n in 9; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DMEDT_ABL=$n -I include -I $C -c $C/axial_bwd.hip -o $C/build/abl/axial_bwd_$n.o &
  done; wait
  for n in 9; do
    objs=$(ls $C/build/*.o | grep -v axial_bwd.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $C/build/abl/axial_bwd_$n.o -o $C/build/abl/libmedt_abl$n.so
  done
  ls -la $C/build/abl/*.so
else
  mkdir -p gpurun_out/r3abl
  python bench.py --roofline-only 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('ABL 0 bwd_core ms', j['roofline']['bwd_core']['launch_ms'])"
  for n in 9; do
    MEDT_LIB_OVERRIDE=$PWD/$C/build/abl/libmedt_abl$n.so python bench.py --roofline-only 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('ABL $n bwd_core ms', j['roofline']['bwd_core']['launch_ms'])"
  done
fi
