#!/bin/bash
# Round 6, call 16: bn_similarity / bn_output (forward) and bn_output (backward) finalised by their consumers (csrc/fin_inline.h)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r6_call16}
rm -rf $O && mkdir -p $O
timeout 900 python -m pytest tests/test_axial_layer_gpu.py -m gpu -q -x 2>&1 | tail -4 | tee $O/layer.txt
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "graphed or deferred or factory_state_train_parity or trajectory or fixture" 2>&1 | tail -4 | tee $O/model.txt
b() { name=$1; shift; echo -n "$name " >> $O/ab.txt; env "$@" timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d.get('fwd_ms_per_image'), d.get('fwd_ms_per_image_bs1'))" >> $O/ab.txt 2>&1; }
b INLINE_OFF MEDT_INLINE_FIN=0
b INLINE_ON A=1
b INLINE_OFF2 MEDT_INLINE_FIN=0
b INLINE_ON2 A=1
for cfg in "medt256 --model MedT --imgsize 256 --batch 2" "gated_f32 --model gatedaxialunet --batch 8"; do
  set -- $cfg; name=$1; shift
  for t in 0 1; do
    echo -n "$name INLINE=$t " >> $O/ab.txt
    MEDT_INLINE_FIN=$t timeout 300 python bench.py "$@" --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" >> $O/ab.txt 2>&1
  done
done
cat $O/ab.txt
bash scripts/r6_trace.sh $(basename $O)/trace
