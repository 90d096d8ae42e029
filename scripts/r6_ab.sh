#!/bin/bash
# Round 6 A/B ON the GPU box: the product library against medical-transformer_amd/libmedt_ab.so (medt_amd.build.build_ab("-DMEDT_AB_..."),
# built before the gpurun call), alternating, on the three BASELINE configurations.  usage: r6_ab.sh <outdir> [quick]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r6_ab}
rm -rf $O && mkdir -p $O
AB=$GRAFT_REPO_ROOT/medical-transformer_amd/libmedt_ab.so
[ -f "$AB" ] || { echo "build libmedt_ab.so first"; exit 1; }
b() { name=$1; shift; echo -n "$name " >> $O/ab.txt; env "$@" timeout 300 python bench.py $CFG --steps 100 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d.get('fwd_ms_per_image'))" >> $O/ab.txt 2>&1; }
for cfg in "--model MedT" "--model gatedaxialunet --batch 8" "--model MedT --imgsize 256 --batch 2"; do
  CFG="$cfg"; echo "# $cfg" >> $O/ab.txt
  b product A=1; b ab MEDT_LIB_OVERRIDE=$AB; b product A=1; b ab MEDT_LIB_OVERRIDE=$AB
  [ "$2" = quick ] && break
done
cat $O/ab.txt
