#!/bin/bash
# Round 6: per-kernel time of the gatedaxialunet bs 8 step with fp32 and with bf16 storage (rocprofv3 --kernel-trace --stats)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6_bf16
rm -rf $O && mkdir -p $O
for dt in f32 bf16; do
  extra=""; [ $dt = bf16 ] && extra="--dtype bf16"
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$dt -- python bench.py --model gatedaxialunet --batch 8 $extra --no-cpu-baseline --no-roofline > $O/$dt.log 2>&1
  T=$(ls -S $(find $O/$dt -name "*kernel_trace.csv") | head -1)
  python scripts/step_timeline.py $T $O/timeline_$dt.json > $O/timeline_$dt.txt 2>&1
  rm -rf $O/$dt
done
python - <<PY
import json
def per(f):
    j=json.load(open(f)); out={}; n=0
    for q,v in j['queues'].items():
        n+=v['launches']
        for k,e in v['kernels'].items():
            o=out.setdefault(k,[0,0.0]); o[0]+=e['launches']; o[1]+=e['busy_us']
    return out,n
a,na=per('$O/timeline_f32.json'); b,nb=per('$O/timeline_bf16.json')
print('launches', na, nb, 'busy', sum(v[1] for v in a.values()), sum(v[1] for v in b.values()))
for k in sorted(set(a)|set(b), key=lambda k:-abs(b.get(k,[0,0])[1]-a.get(k,[0,0])[1])):
    x=a.get(k,[0,0]); y=b.get(k,[0,0])
    if abs(x[1]-y[1])>4: print(f"{k:45s} {x[0]:3d} {x[1]:8.1f} -> {y[0]:3d} {y[1]:8.1f}  ({y[1]-x[1]:+.1f})")
PY
for dt in f32 bf16 f32 bf16; do
  extra=""; [ $dt = bf16 ] && extra="--dtype bf16"
  echo -n "gated $dt " >> $O/ab.txt
  timeout 300 python bench.py --model gatedaxialunet --batch 8 $extra --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" >> $O/ab.txt
done
cat $O/ab.txt
timeout 900 python -m pytest tests/test_axial_layer_gpu.py tests/test_model_gpu.py -m gpu -q -x -k "bf16" 2>&1 | tail -3
