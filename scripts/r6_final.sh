#!/bin/bash
# Round 6 collection ON the GPU box (gpurun): full GPU suite + parity report, smoke, the bench lines of every BASELINE configuration, kernel
# statistics + chains of the step, the remaining library switches all OFF at once (fallback paths), SQ counters of the step's kernels.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r6_final}
rm -rf $O && mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -E "product error|label map|rel err|worst gradient|trajectory|top-5|passed|failed|FAILED|Error|eval forward|factory state|assert" > $O/parity_report.txt
tail -3 $O/parity_report.txt
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -3 $O/smoke.log
timeout 600 python bench.py 2>$O/bench.err | tail -1 > $O/bench_line.json
python -c "import json; j=json.load(open('$O/bench_line.json')); print('step', j['ms_per_step'], j['value'], 'fwd', j['fwd_ms_per_image'], j['fwd_ms_per_image_bs1'], j.get('fwd_ms_per_image_eager'), 'roof', j['roofline']['frac'], j['roofline']['bwd_core']['frac'], j['roofline']['in_model_shape'], 'cpu', j['cpu_baseline']['value'], j['vs_cpu_baseline'])"
for cfg in "medt256 --model MedT --imgsize 256 --batch 2" "gated_f32 --model gatedaxialunet --batch 8" "gated_bf16 --model gatedaxialunet --batch 8 --dtype bf16"; do
  set -- $cfg; name=$1; shift
  timeout 300 python bench.py "$@" --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_line_$name.json
  python -c "import json; j=json.load(open('$O/bench_line_$name.json')); print('$name', j['ms_per_step'], j['value'], j.get('fwd_ms_per_image'), j.get('fwd_ms_per_image_bs1'))"
done
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python bench.py --no-cpu-baseline --no-roofline > $O/bench_prof.log 2>&1
T=$(ls -S $(find $O/bench -name "*kernel_trace.csv") | head -1)
python scripts/step_chains.py $T $O/step_chains.json 12 > $O/step_chains.txt 2>&1
python scripts/step_timeline.py $T $O/step_timeline.json > $O/step_timeline.txt 2>&1
cp $(ls -S $(find $O/bench -name "*kernel_stats.csv") | head -1) $O/bench_kernel_stats.csv; rm -rf $O/bench
grep -E "^local|^global" $O/step_chains.txt
# fallback paths: every remaining library switch that selects a path, OFF at once
( export MEDT_INLINE_FIN=0 MEDT_CONV_STEM7=0 MEDT_CONV_THIN=0 MEDT_BWD_WIDE=0 MEDT_TWO_BUCKETS=0 MEDT_BLOCK_BWD=0 MEDT_BLOCK8=0 MEDT_BLOCK_S2=0 MEDT_F4R_VEC=0
  timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py tests/test_block_gpu.py tests/test_infer_gpu.py tests/test_axial_layer_gpu.py -m gpu -q 2>&1 | tail -4 | tee $O/fallbacks.txt
  timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('all path switches off:', d['ms_per_step'], d['value'])" | tee -a $O/fallbacks.txt )
C="python bench.py --no-cpu-baseline --no-roofline --steps 3 --warmup 1"
MEDT_BENCH_WINDOWS=1 timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/pmc1 -- $C > $O/pmc1.log 2>&1
MEDT_BENCH_WINDOWS=1 timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d $O/pmc2 -- $C > $O/pmc2.log 2>&1
O=$O python - <<'PY'
import collections, csv, glob, json, os, re
O = os.environ["O"]
out = {}
for d in ("pmc1", "pmc2"):
    fs = glob.glob(f"{O}/{d}/**/*_counter_collection.csv", recursive=True)
    if not fs:
        out[d + "_error"] = "no counter file (see %s.log)" % d
        continue
    f = max(fs, key=os.path.getsize)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("medt::", "").replace("(anonymous namespace)::", "").replace("void ", ""))
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, dd in agg.items():
        e = out.setdefault(k, {})
        for c, v in dd.items():
            e[c] = sum(v) / len(v)
        e["launches_" + d] = len(next(iter(dd.values())))
for k, e in out.items():
    if isinstance(e, dict) and e.get("SQ_WAVE_CYCLES"):
        wc = e["SQ_WAVE_CYCLES"]
        e["derived"] = {"mfma_busy_frac_of_wave_cycles_at_1_wave_per_simd": round(e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 4.0 / wc, 3),
                        "wait_any_frac": round(e.get("SQ_WAIT_ANY", 0.0) / wc, 3), "wait_inst_frac": round(e.get("SQ_WAIT_INST_ANY", 0.0) / wc, 3),
                        "salu_per_wave": round(e.get("SQ_INSTS_SALU", 0.0) / max(e.get("SQ_WAVES", 1.0), 1.0), 1),
                        "valu_per_wave": round(e.get("SQ_INSTS_VALU", 0.0) / max(e.get("SQ_WAVES", 1.0), 1.0), 1)}
json.dump({"source": "rocprofv3 --pmc (two passes, --kernel-trace only; scripts/r6_final.sh) on `python bench.py --no-cpu-baseline --no-roofline --steps 3 --warmup 1` (MedT 128, bs 4); per-launch averages; counters summed over the chip", "kernels": out}, open(f"{O}/step_pmc.json", "w"), indent=0)
print("pmc kernels:", len(out))
PY
rm -rf $O/pmc1 $O/pmc2
du -sh $O
