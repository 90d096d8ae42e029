cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/profiles_raw; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_block_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "conv_block or block or MedT_S256 or medt_256 or MedT_S128_N4 or deferred" 2>&1 | tail -3
timeout 300 python bench.py --model MedT --imgsize 256 --batch 2 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_line_medt256.json
timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_line_quick.json
python -c "import json; [print(f, round(json.load(open('$O/'+f))['ms_per_step'],4)) for f in ('bench_line_medt256.json','bench_line_quick.json')]"
