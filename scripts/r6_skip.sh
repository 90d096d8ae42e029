# round 6, TIMING EXPERIMENT: the step with one kernel family removed per run (MEDT_SKIP, medt_common.h: abl_skip) -- the time that
# disappears is the family's share of the critical path (results are garbage; run ON the GPU box through gpurun)
# The product library has no MEDT_SKIP: medical-transformer_amd/libmedt_ablate.so (python -c "from medt_amd import build; build.build_ablate()",
# built before the gpurun call: it travels with the snapshot) is loaded through MEDT_LIB_OVERRIDE.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export MEDT_LIB_OVERRIDE=$GRAFT_REPO_ROOT/medical-transformer_amd/libmedt_ablate.so
[ -f "$MEDT_LIB_OVERRIDE" ] || { echo "build libmedt_ablate.so first"; exit 1; }
O=gpurun_out/${AB_OUT:-r6_skip}
rm -rf $O && mkdir -p $O
for fam in NONE sweep attn_fwd wopos_fwd wopos_bwd block_fwd block_bwd flush "sweep,attn_fwd" "wopos_fwd,wopos_bwd,block_fwd,block_bwd" conv1_fwd_g conv1_dgrad_g rows16 conv_thin NONE; do
  echo -n "$fam " >> $O/skip.txt
  MEDT_SKIP=$fam timeout 120 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step'],4))" >> $O/skip.txt 2>&1
done
cat $O/skip.txt
