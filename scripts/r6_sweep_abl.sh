#!/bin/bash
# Round 6 TIMING EXPERIMENT on the in-model single sweep: library variants that leave one piece of the kernel out (-DMEDT_ABL=n: 6 no row loop,
# 10 no consumer-side bn_output finalisation, 11 no table-gradient / Gram epilogue; results are garbage), per-instance launch time inside the step.
#   here:        bash scripts/r6_sweep_abl.sh build      (libmedt_abl<n>.so next to the product library: they travel with the snapshot)
#   on the box:  bash scripts/r6_sweep_abl.sh run
cd "$(dirname "$0")/.."
C=medical-transformer_amd/csrc
NS="${NS:-6 10 11}"
if [ "$1" = build ]; then
  mkdir -p $C/build/abl
  for n in $NS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DMEDT_ABL=$n -I include -I $C -c $C/axial_bwd.hip -o $C/build/abl/axial_bwd_$n.o 2>/dev/null &
  done; wait
  for n in $NS; do
    objs=$(ls $C/build/*.o | grep -v "axial_bwd.o")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $C/build/abl/axial_bwd_$n.o -o medical-transformer_amd/libmedt_abl$n.so
  done
  ls -la medical-transformer_amd/libmedt_abl*.so
else
  cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
  O=gpurun_out/r6_sweep_abl; rm -rf $O; mkdir -p $O
  for n in 0 $NS; do
    [ $n = 0 ] && unset MEDT_LIB_OVERRIDE || export MEDT_LIB_OVERRIDE=$GRAFT_REPO_ROOT/medical-transformer_amd/libmedt_abl$n.so
    for cfg in "medt --model MedT" ${ABL_GATED:+"gated --model gatedaxialunet --batch 8"}; do
      set -- $cfg; name=$1; shift
      timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python bench.py "$@" --no-cpu-baseline --no-roofline --steps 20 --warmup 5 > $O/log_${n}_$name.txt 2>&1
      S=$(ls -S $(find $O/p -name "*kernel_stats.csv") | head -1)
      echo "== ABL $n $name" >> $O/sweep.txt
      grep -E "attn_bwd_sweep|attn_bwd_fix|attn_bwd_relfix" $S | python -c "
import sys,csv,re
for r in csv.reader(sys.stdin):
    n=re.sub(r'\(.*','',r[0].replace('medt::','').replace('(anonymous namespace)::','').replace('void ',''))
    print(f'  {n:50s} calls {r[1]:>5s} avg {float(r[3])/1e3:7.1f} us')" >> $O/sweep.txt
      rm -rf $O/p
    done
  done
  cat $O/sweep.txt
fi
