cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in "DEFAULT:" "NO_BLOCK:MEDT_BLOCK_FUSED=0" "NO_EARLY_FIN:MEDT_EARLY_FIN=0" "ONE_STREAM:MEDT_TWO_STREAMS=0"; do
  name=${v%%:*}; envs=${v#*:}
  echo -n "$name "; env $envs timeout 100 python scripts/eval_fwd_time.py 2>&1 | tail -1
done
