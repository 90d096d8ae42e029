cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3i; mkdir -p $O
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('sweep step ms', j['ms_per_step'])"; done
for i in 1 2; do MEDT_BWD_SWEEP=0 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('two-pass step ms', j['ms_per_step'])"; done
rocm-smi --showclocks 2>/dev/null | head -20
