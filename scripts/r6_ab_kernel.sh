#!/bin/bash
# Round 6 A/B ON the GPU box with per-kernel times: scripts/r6_ab.sh (step times, product vs libmedt_ab.so) + rocprofv3 averages of the kernels
# matching $2 (a grep -E pattern) in the replayed step with each library.  usage: r6_ab_kernel.sh <outdir> <pattern> [quick]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r6_abk}; PAT=${2:-wgrad}
bash scripts/r6_ab.sh ${1:-r6_abk} $3 > /dev/null 2>&1
AB=$GRAFT_REPO_ROOT/medical-transformer_amd/libmedt_ab.so
for side in product ab; do
  [ $side = ab ] && export MEDT_LIB_OVERRIDE=$AB || unset MEDT_LIB_OVERRIDE
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 5 > $O/log_$side.txt 2>&1
  S=$(ls -S $(find $O/p -name "*kernel_stats.csv") | head -1)
  echo "== $side" >> $O/kernels.txt
  grep -E "$PAT" $S | python -c "
import sys,csv,re
for r in csv.reader(sys.stdin):
    n=re.sub(r'\(.*','',r[0].replace('medt::','').replace('(anonymous namespace)::','').replace('void ',''))
    print(f'  {n:55s} calls {r[1]:>5s} avg {float(r[3])/1e3:7.1f} us')" >> $O/kernels.txt
  rm -rf $O/p
done
cat $O/ab.txt $O/kernels.txt
