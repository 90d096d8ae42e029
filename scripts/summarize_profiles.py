#!/usr/bin/env python
"""Turn gpurun_out/profiles_raw/ (written by scripts/collect_profiles.sh on the GPU box) into the tracked summaries
under profiles/: kernel statistics of the bench commands, the HBM traffic of the roofline kernels (FETCH_SIZE x2 on
gfx950, MI355X_MICROARCH.md) and the SQ counters of the attention forward kernel."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RAW = os.path.join(ROOT, "gpurun_out", "profiles_raw")
OUT = os.path.join(ROOT, "profiles")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r01"


def one(pattern):
    fs = sorted(glob.glob(os.path.join(RAW, pattern)), key=os.path.getmtime)    # gpurun merges runs: newest wins
    if not fs:
        raise SystemExit("missing " + pattern)
    return fs[-1]


def counters(folder):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(one(folder + "/*/*_counter_collection.csv"))):
        agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} | {"launches": len(next(iter(d.values())))} for k, d in agg.items()}


shutil.copy(one("bench/*/*_kernel_stats.csv"), os.path.join(OUT, TAG + "_bench_kernel_stats.csv"))
shutil.copy(one("roofline/*/*_kernel_stats.csv"), os.path.join(OUT, TAG + "_roofline_kernel_stats.csv"))
shutil.copy(os.path.join(RAW, "bench_line.json"), os.path.join(OUT, TAG + "_bench_line.json"))

fetch, write = counters("pmc_fetch"), counters("pmc_write")
main = [k for k in fetch if "attn_fwd" in k and fetch[k]["FETCH_SIZE"] > 1000][0]
kern = {}
for k in fetch:
    if "attn_fwd" in k or "logit_stats" in k:
        kern[k] = {"FETCH_SIZE_KB_avg": fetch[k]["FETCH_SIZE"], "WRITE_SIZE_KB_avg": write.get(k, {}).get("WRITE_SIZE"),
                   "launches": fetch[k]["launches"]}
traffic = int((2 * fetch[main]["FETCH_SIZE"] + write[main]["WRITE_SIZE"]) * 1024)
json.dump({
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python bench.py --roofline-only` (scripts/collect_profiles.sh)",
    "correction": "gfx950: FETCH_SIZE reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section) -> x2; WRITE_SIZE as reported; unit KiB",
    "kernels": kern,
    "attn_fwd_kernel": main,
    "attn_fwd_bytes_per_launch": traffic,
    "attn_fwd_algorithmic_bytes": 268435456,
    "note": "traffic = qkv read once (134 MB) + sv|sve written once (134 MB) + row log-sum-exp (33.5 MB, excluded from the algorithmic figure by SURVEY.md 8d)",
}, open(os.path.join(OUT, "roofline_traffic.json"), "w"), indent=1)

sq = counters("pmc_sq1")
for k, d in counters("pmc_sq2").items():
    sq.setdefault(k, {}).update(d)
m = sq[main]
stats = {r["Name"].split("(")[0].replace("void ", ""): r for r in csv.DictReader(open(one("roofline/*/*_kernel_stats.csv")))}
avg_ns = float(stats[main]["AverageNs"])
cus, simds = 256, 1024
cycles = m["GRBM_GUI_ACTIVE"] / 8.0                     # summed over the 8 XCDs
derived = {
    "kernel": main,
    "avg_launch_us_unprofiled_stats_run": avg_ns / 1e3,
    "shader_cycles_per_launch": cycles,
    "effective_clock_GHz_in_the_counter_pass": cycles / (m.get("_dur_ns", avg_ns)),
    "valu_busy_frac": 4.0 * m["SQ_ACTIVE_INST_VALU"] / simds / cycles,          # SQ_ACTIVE_INST_* count quad-cycles
    "lds_array_busy_frac": m["SQ_LDS_IDX_ACTIVE"] / cus / cycles,
    "lds_bank_conflict_cycles": m["SQ_LDS_BANK_CONFLICT"],
    "valu_instructions_per_launch": m["SQ_INSTS_VALU"],
    "lds_instructions_per_launch": m["SQ_INSTS_LDS"],
    "waves_resident_per_simd_avg": 4.0 * m["SQ_WAVE_CYCLES"] / simds / cycles,
    "wait_any_frac_of_wave_cycles": m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"],
    "wait_inst_any_frac_of_wave_cycles": m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"],
}
json.dump({"source": "rocprofv3 --pmc (two SQ passes, scripts/collect_profiles.sh) on `python bench.py --roofline-only`",
           "units": "SQ_ACTIVE_INST_*/SQ_WAVE_CYCLES/SQ_WAIT_* in quad-cycles summed over the chip; GRBM_GUI_ACTIVE summed over 8 XCDs",
           "raw_avg_per_launch": {k: v for k, v in sq.items() if "attn_fwd" in k or "logit_stats" in k},
           "derived": derived}, open(os.path.join(OUT, TAG + "_attn_fwd_pmc.json"), "w"), indent=1)
print(json.dumps(derived, indent=1))
print("traffic bytes/launch", traffic)
