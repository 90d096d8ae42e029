#!/bin/bash
# Round 5 final collection ON the GPU box (gpurun): full GPU suite + parity report, smoke, the bench lines of every BASELINE
# configuration, kernel statistics + chains of the step, SQ counters of the step's kernels (two --pmc passes, --kernel-trace only).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_final
rm -rf $O && mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -E "product error|label map|rel err|worst gradient|trajectory|top-5|passed|failed|FAILED|Error|eval forward|factory state|assert" > $O/parity_report.txt
tail -3 $O/parity_report.txt
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -3 $O/smoke.log
timeout 600 python bench.py 2>$O/bench.err | tail -1 > $O/bench_line.json
python -c "import json; j=json.load(open('$O/bench_line.json')); print('step', j['ms_per_step'], j['value'], 'fwd', j['fwd_ms_per_image'], j['fwd_ms_per_image_bs1'], j['fwd_ms_per_image_eager'], 'roof', j['roofline']['frac'], j['roofline']['bwd_core']['frac'], j['roofline']['in_model_shape'], 'cpu', j['cpu_baseline']['value'], j['vs_cpu_baseline'])"
for cfg in "medt256 --model MedT --imgsize 256 --batch 2" "gated_f32 --model gatedaxialunet --batch 8" "gated_bf16 --model gatedaxialunet --batch 8 --dtype bf16"; do
  set -- $cfg; name=$1; shift
  timeout 300 python bench.py "$@" --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_line_$name.json
  python -c "import json; j=json.load(open('$O/bench_line_$name.json')); print('$name', j['ms_per_step'], j['value'], j.get('fwd_ms_per_image'), j.get('fwd_ms_per_image_bs1'))"
done
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python bench.py --no-cpu-baseline --no-roofline > $O/bench_prof.log 2>&1
T=$(ls -S $O/bench/*/*_kernel_trace.csv | head -1)
python scripts/step_chains.py $T $O/step_chains.json 12 > $O/step_chains.txt 2>&1
cp $(ls -S $O/bench/*/*_kernel_stats.csv | head -1) $O/bench_kernel_stats.csv; rm -rf $O/bench
grep -E "^local|^global" $O/step_chains.txt
C="python bench.py --no-cpu-baseline --no-roofline --steps 3 --warmup 1"
MEDT_BENCH_WINDOWS=1 timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/pmc1 -- $C > $O/pmc1.log 2>&1
MEDT_BENCH_WINDOWS=1 timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d $O/pmc2 -- $C > $O/pmc2.log 2>&1
python - <<'PY'
import collections, csv, glob, json, os, re
O = "gpurun_out/r5_final"
out = {}
for d in ("pmc1", "pmc2"):
    fs = glob.glob(f"{O}/{d}/*/*_counter_collection.csv")
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
json.dump({"source": "rocprofv3 --pmc (two passes, --kernel-trace only; scripts/r5_final.sh) on `python bench.py --no-cpu-baseline --no-roofline --steps 3 --warmup 1` (MedT 128, bs 4); per-launch averages; counters summed over the chip", "kernels": out}, open(f"{O}/step_pmc.json", "w"), indent=0)
print("pmc kernels:", len(out))
for k in out:
    if any(t in k for t in ("wgrad_mfma_grouped", "rows16", "stem7")) and isinstance(out[k], dict) and "derived" in out[k]:
        print(k[:60], out[k]["derived"])
PY
rm -rf $O/pmc1 $O/pmc2
du -sh $O
