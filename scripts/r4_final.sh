#!/bin/bash
# Round-4 final lines ON the GPU box (gpurun): the full bench line, kernel statistics + chain summary of the step, and the SQ
# counters of the step's kernels reduced to per-kernel averages on the box (the raw counter tables exceed gpurun's 64 MiB merge limit).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final
rm -rf gpurun_out/* && mkdir -p $O
timeout 500 python bench.py 2>$O/bench.err | tail -1 > $O/bench_line.json
python -c "import json; j=json.load(open('$O/bench_line.json')); print('step', j['ms_per_step'], 'fwd', j['fwd_ms_per_image'], j.get('box_probe'))"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python bench.py --no-cpu-baseline --no-roofline > $O/bench_prof.log 2>&1
T=$(ls -S $O/bench/*/*_kernel_trace.csv | head -1)
python scripts/step_chains.py $T $O/step_chains.json 12 > $O/step_chains.txt 2>&1
python scripts/step_timeline.py $T $O/step_timeline.json > /dev/null 2>&1
cp $(ls -S $O/bench/*/*_kernel_stats.csv | head -1) $O/bench_kernel_stats.csv
rm -rf $O/bench
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc1 -- python bench.py --no-cpu-baseline --no-roofline --steps 5 --warmup 2 > $O/pmc1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc2 -- python bench.py --no-cpu-baseline --no-roofline --steps 5 --warmup 2 > $O/pmc2.log 2>&1
python - <<'PY'
import collections, csv, glob, json, os, re
O = "gpurun_out/final"
out = {}
for d in ("pmc1", "pmc2"):
    fs = glob.glob(f"{O}/{d}/*/*_counter_collection.csv")
    if not fs:
        out[d + "_error"] = "no counter file"
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
json.dump(out, open(f"{O}/step_pmc.json", "w"), indent=0)
print("pmc kernels:", len(out))
PY
rm -rf $O/pmc1 $O/pmc2
du -sh gpurun_out; head -30 $O/step_chains.txt | cut -c1-100
