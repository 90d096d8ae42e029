#!/bin/bash
# Round 5: SQ counters of the attention kernels on the roofline shapes after the forward main pass was reworked, plus the HBM
# traffic passes behind profiles/roofline_traffic.json (separate --pmc runs with --kernel-trace only; MI355X_MICROARCH.md).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_fwd_pmc
rm -rf $O && mkdir -p $O
R="python bench.py --roofline-only"
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq1 -- $R > $O/pmc_sq1.log 2>&1
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq2 -- $R > $O/pmc_sq2.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $R > $O/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $R > $O/pmc_write.log 2>&1
timeout 200 $R 2>/dev/null | grep '^{"roofline' | tail -1 > $O/roofline_only.json
python scripts/r5_traffic.py $O/pmc_fetch $O/pmc_write $O/roofline_only.json "$(cat .commit_stamp 2>/dev/null)" > $O/roofline_traffic.json 2>$O/traffic.err; tail -2 $O/traffic.err
python - <<'PY'
import collections, csv, glob, json, os, re
O = "gpurun_out/r5_fwd_pmc"
out = {}
for d in ("pmc_sq1", "pmc_sq2"):
    fs = glob.glob(f"{O}/{d}/*/*_counter_collection.csv")
    if not fs:
        out[d + "_error"] = "no counter file"
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(max(fs, key=os.path.getsize))):
        k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", ""))
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, dd in agg.items():
        if "attn" not in k and "sim_stats" not in k:
            continue
        e = out.setdefault(k, {})
        for c, v in dd.items():
            e[c] = sum(v) / len(v)
        e["launches"] = len(next(iter(dd.values())))
for k, e in out.items():
    if not isinstance(e, dict) or "SQ_WAVE_CYCLES" not in e:
        continue
    e["derived"] = {"valu_busy_frac_of_busy": e["SQ_ACTIVE_INST_VALU"] / (4 * e["SQ_BUSY_CYCLES"]) if e.get("SQ_BUSY_CYCLES") else None,
                    "wait_frac_of_wave_cycles": e["SQ_WAIT_INST_ANY"] / e["SQ_WAVE_CYCLES"],
                    "lds_conflict_frac": e.get("SQ_LDS_BANK_CONFLICT", 0) / e["SQ_LDS_IDX_ACTIVE"] if e.get("SQ_LDS_IDX_ACTIVE") else None}
json.dump(out, open(O + "/attn_pmc.json", "w"), indent=1)
for k, e in out.items():
    if "fwd4r" in k and isinstance(e, dict):
        print(k, json.dumps(e)[:900])
PY
rm -rf $O/pmc_sq1 $O/pmc_sq2 $O/pmc_fetch $O/pmc_write
cat $O/roofline_traffic.json | head -c 1500
