# round 4, TIMING EXPERIMENT: the step with one kernel family removed per run (MEDT_SKIP, medt_common.h: abl_skip) -- the time that
# disappears is the family's share of the critical path (results are garbage; run ON the GPU box through gpurun)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${AB_OUT:-r4_skip}
rm -rf $O && mkdir -p $O
for fam in NONE wopos_fwd block_fwd wopos_bwd conv_small_fwd bn_dgrad qkv_dgrad_l rows16 wgrad_mfma_l conv3_fwd_l conv3_dgrad_l conv3_fwd_g conv3_dgrad_g conv7_fwd_l conv7_fwd_g conv1_fwd_g conv1_dgrad_g qkv_dgrad_g sweep attn_fwd flush "sweep,wopos_bwd" "sweep,wopos_bwd,bn_dgrad,qkv_dgrad_l" $EXTRA_FAMS NONE; do
  echo -n "$fam " >> $O/skip.txt
  MEDT_SKIP=$fam timeout 120 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step'],4))" >> $O/skip.txt 2>&1
done
cat $O/skip.txt
