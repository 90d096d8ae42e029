#!/bin/bash
# PMC of the step's own kernels (not the roofline shape): matrix-pipe busy, issue stalls, memory waits per kernel.
# One counter group per pass, --kernel-trace only (MI355X_MICROARCH.md); summarised by the inline python below into
# gpurun_out/profiles_raw/step_pmc.json.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/profiles_raw
mkdir -p $O; rm -rf $O/pmc_step1 $O/pmc_step2
C="python bench.py --no-cpu-baseline --no-roofline --steps 3 --warmup 1"
MEDT_BENCH_WINDOWS=1 timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/pmc_step1 -- $C > $O/pmc_step1.log 2>&1
MEDT_BENCH_WINDOWS=1 timeout 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/pmc_step2 -- $C > $O/pmc_step2.log 2>&1
python - <<'PY'
import collections, csv, glob, json, os
def counters(folder):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    f = max(glob.glob("gpurun_out/profiles_raw/" + folder + "/*/*_counter_collection.csv"), key=os.path.getsize)   # (the box probe runs as a child process and leaves its own small file)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} | {"launches": len(next(iter(d.values())))} for k, d in agg.items()}
a, b = counters("pmc_step1"), counters("pmc_step2")
out = {}
for k, m in a.items():
    if not k.startswith("medt::") or m.get("SQ_WAVE_CYCLES", 0) < 1e4:
        continue
    m = dict(m); m.update({kk: vv for kk, vv in b.get(k, {}).items() if kk != "launches"})
    wc = m["SQ_WAVE_CYCLES"]
    m["mfma_busy_frac_of_wave_cycles_at_1_wave_per_simd"] = round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 4.0 / wc, 3)
    m["wait_any_frac"] = round(m.get("SQ_WAIT_ANY", 0.0) / wc, 3)
    m["wait_inst_any_frac"] = round(m.get("SQ_WAIT_INST_ANY", 0.0) / wc, 3)
    m["active_inst_frac"] = round(m.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 3)
    if "SQ_ACTIVE_INST_VALU" in m: m["valu_active_frac"] = round(m["SQ_ACTIVE_INST_VALU"] / wc, 3)
    if m.get("SQ_LDS_IDX_ACTIVE"): m["lds_bank_conflict_frac_of_lds_active"] = round(m.get("SQ_LDS_BANK_CONFLICT", 0.0) / m["SQ_LDS_IDX_ACTIVE"], 3)
    out[k] = m
json.dump({"source": "rocprofv3 --pmc (two passes, scripts/r3_step_pmc.sh) --kernel-trace on `python bench.py --no-cpu-baseline --no-roofline --steps 3 --warmup 1` (MedT 128, bs 4); per-launch averages over all launches of the kernel in the run. SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs",
           "kernels": dict(sorted(out.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"] * kv[1]["launches"]))},
          open("gpurun_out/profiles_raw/step_pmc.json", "w"), indent=1)
for k, m in list(out.items())[:0]: print(k)
PY
find $O/pmc_step1 $O/pmc_step2 -name "*kernel_trace.csv" -delete
