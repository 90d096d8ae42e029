cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/profiles_raw; mkdir -p $O
timeout 500 python bench.py 2>$O/bench.err | tail -1 > $O/bench_line.json
python -c "import json; j=json.load(open('$O/bench_line.json')); print('step', j['ms_per_step'], j['value'], j.get('box_probe'))"
bash scripts/r3_step_pmc.sh
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -3
timeout 120 python __graft_entry__.py --smoke 2>&1 | tail -2
