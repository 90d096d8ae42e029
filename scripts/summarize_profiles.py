#!/usr/bin/env python
"""Turn gpurun_out/profiles_raw/ (written by scripts/collect_profiles.sh on the GPU box) into the tracked summaries
under profiles/: kernel statistics of the bench commands, launch counts of one replayed step, the HBM traffic of the
roofline kernels (FETCH_SIZE x2 on gfx950, MI355X_MICROARCH.md), the SQ counters of the attention kernels, the A/B
tables and the parity report.  usage: summarize_profiles.py r02"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RAW = os.path.join(ROOT, "gpurun_out", "profiles_raw")
OUT = os.path.join(ROOT, "profiles")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r02"


def one(pattern, required=True):
    fs = sorted(glob.glob(os.path.join(RAW, pattern)), key=os.path.getmtime)    # gpurun merges runs: newest wins
    if not fs:
        if required:
            raise SystemExit("missing " + pattern)
        return None
    # ... and within the newest run the largest file: bench.py's box probe runs as a child process and leaves its own
    # (tiny) trace / statistics files next to the bench's
    newest = os.path.getmtime(fs[-1])
    return max((f for f in fs if newest - os.path.getmtime(f) < 120), key=os.path.getsize)


def short(name):
    return name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")


def counters(folder):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(one(folder + "/*/*_counter_collection.csv"))):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} | {"launches": len(next(iter(d.values())))} for k, d in agg.items()}


def stats(folder):
    return {short(r["Name"]): r for r in csv.DictReader(open(one(folder + "/*/*_kernel_stats.csv")))}


def jline(name):
    p = os.path.join(RAW, name)
    try:
        return json.loads(open(p).read().strip().splitlines()[-1])
    except Exception:
        return None


# ---- bench lines ---------------------------------------------------------------------------------------------
shutil.copy(one("bench/*/*_kernel_stats.csv"), os.path.join(OUT, TAG + "_bench_kernel_stats.csv"))
shutil.copy(one("roofline/*/*_kernel_stats.csv"), os.path.join(OUT, TAG + "_roofline_kernel_stats.csv"))
shutil.copy(one("roofline_h/*/*_kernel_stats.csv"), os.path.join(OUT, TAG + "_roofline_h_kernel_stats.csv"))
shutil.copy(one("roofline_bf16/*/*_kernel_stats.csv"), os.path.join(OUT, TAG + "_roofline_bf16_kernel_stats.csv"))
shutil.copy(os.path.join(RAW, "bench_line.json"), os.path.join(OUT, TAG + "_bench_line.json"))
other = {k: jline(f) for k, f in (("gatedaxialunet_bs8_bf16 (BASELINE configs[1])", "bench_line_gated_bf16.json"),
                                  ("gatedaxialunet_bs8_f32", "bench_line_gated_f32.json"),
                                  ("MedT_256_bs2 (configs[4] per GPU)", "bench_line_medt256.json"))}
json.dump(other, open(os.path.join(OUT, TAG + "_bench_other_configs.json"), "w"), indent=1)

# ---- launches of one replayed training step (between two adam_step launches late in the trace) ---------------------
tr = one("bench/*/*_kernel_trace.csv", required=False)
if tr:
    rows = sorted(csv.DictReader(open(tr)), key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"] for r in rows]
    ad = [i for i, n in enumerate(names) if "adam_step_kernel" in n]
    step = rows[ad[-3] + 1:ad[-2] + 1]
    cnt, busy = collections.Counter(), collections.Counter()
    for r in step:
        k = re.sub(r"<.*", "", short(r["Kernel_Name"]))
        cnt[k] += 1
        busy[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    json.dump({"source": "rocprofv3 --kernel-trace on `python bench.py --no-cpu-baseline --no-roofline`: the launches between "
                         "two consecutive adam_step kernels of the replayed hipGraph (the profiler serialises the two "
                         "streams, so busy time is per kernel, not wall time)",
               "launches_per_step": len(step),
               "kernels": {k: {"launches": cnt[k], "busy_us": round(busy[k] / 1e3, 1)} for k, _ in busy.most_common()}},
              open(os.path.join(OUT, TAG + "_step_launches.json"), "w"), indent=1)
    print("launches per step:", len(step))

# ---- roofline kernels: durations, traffic ------------------------------------------------------------------------
fetch, write = counters("pmc_fetch"), counters("pmc_write")
rs, rh, rb = stats("roofline"), stats("roofline_h"), stats("roofline_bf16")
roof = jline("bench_line.json")["roofline"]
alg = {"attn_fwd": roof["shape"]["bytes_per_launch"], "sim_stats": roof["stats_kernel"]["bytes_per_launch"],
       "attn_bwd": roof["bwd_core"]["bytes_per_launch"]}


def rows_for(st, tag):
    out = {}
    for k, r in st.items():
        if any(t in k for t in ("attn_fwd", "sim_stats", "attn_bwd", "sim_tables")):
            out[k] = {"avg_us": float(r["AverageNs"]) / 1e3, "calls": int(r["Calls"])}
            for a, nbytes in alg.items():
                if a in k and float(r["AverageNs"]) > 8000:
                    b = nbytes // 2 if tag == "bf16" else nbytes
                    if a == "attn_bwd":
                        continue
                    out[k]["algorithmic_GBps"] = b / float(r["AverageNs"])
                    out[k]["frac_of_8TBps"] = b / float(r["AverageNs"]) / 8000.0
    return out


kern = {}
for k in fetch:
    if any(t in k for t in ("attn_fwd", "sim_stats", "attn_bwd", "sim_bwd_finalize")):
        kern[k] = {"FETCH_SIZE_KB_avg": fetch[k]["FETCH_SIZE"], "WRITE_SIZE_KB_avg": write.get(k, {}).get("WRITE_SIZE"),
                   "launches": fetch[k]["launches"],
                   "hbm_bytes_per_launch": int((2 * fetch[k]["FETCH_SIZE"] + (write.get(k, {}).get("WRITE_SIZE") or 0)) * 1024)}
main = max((k for k in kern if "attn_fwd" in k), key=lambda k: kern[k]["FETCH_SIZE_KB_avg"])
# backward core = the single sweep (frozen gates: the <.., false> instance) + its fix / relfix / finalize launches
bwd_parts = {}
for k in kern:
    if "attn_bwd_sweep" in k and "false" in k:
        bwd_parts["sweep"] = k
    elif "attn_bwd_fix" in k:
        bwd_parts["fix"] = k
    elif "attn_bwd_relfix" in k:
        bwd_parts["relfix"] = k
    elif "sim_bwd_finalize" in k:
        bwd_parts["finalize"] = k
bwd_bytes = sum(kern[k]["hbm_bytes_per_launch"] for k in bwd_parts.values()) if "sweep" in bwd_parts else None
json.dump({
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python bench.py --roofline-only` (scripts/collect_profiles.sh)",
    "correction": "gfx950: FETCH_SIZE reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section) -> x2; WRITE_SIZE as reported; unit KiB",
    "kernels": kern,
    "attn_fwd_kernel": main,
    "attn_fwd_bytes_per_launch": kern[main]["hbm_bytes_per_launch"],
    "attn_fwd_algorithmic_bytes": alg["attn_fwd"],
    "attn_bwd_kernels": bwd_parts,
    "attn_bwd_bytes_per_launch": bwd_bytes,
    "attn_bwd_algorithmic_bytes": alg["attn_bwd"],
    "note": "attn_fwd traffic = qkv read once + sv|sve written once + row log-sum-exp (excluded from the algorithmic figure by SURVEY.md 8d); "
            "sim_stats reads the q,k half of qkv once and writes a few KB of partials",
}, open(os.path.join(OUT, "roofline_traffic.json"), "w"), indent=1)
json.dump({"source": "rocprofv3 --kernel-trace --stats on `python bench.py --roofline-only` (width axis), MEDT_ROOF_AXIS=h (height "
                     "axis) and --dtype bf16; shape C=16 G=8 L=64 B*=16384 (SURVEY.md 8d)",
           "algorithmic_bytes_per_launch_f32": alg,
           "width_axis_f32": rows_for(rs, "f32"), "height_axis_f32": rows_for(rh, "f32"), "width_axis_bf16": rows_for(rb, "bf16"),
           "bench_line_roofline": roof}, open(os.path.join(OUT, TAG + "_roofline_kernels.json"), "w"), indent=1)

# ---- SQ counters of the attention kernels ------------------------------------------------------------------------------
sq = counters("pmc_sq1")
for k, d in counters("pmc_sq2").items():
    sq.setdefault(k, {}).update(d)
cus, simds = 256, 1024
derived = {}
for k, m in sq.items():
    if not any(t in k for t in ("attn_fwd", "sim_stats", "attn_bwd")) or "GRBM_GUI_ACTIVE" not in m or m["GRBM_GUI_ACTIVE"] < 1e4:
        continue
    cycles = m["GRBM_GUI_ACTIVE"] / 8.0                     # summed over the 8 XCDs
    derived[k] = {
        "shader_cycles_per_launch": cycles,
        "valu_busy_frac": 4.0 * m["SQ_ACTIVE_INST_VALU"] / simds / cycles,          # SQ_ACTIVE_INST_* count quad-cycles
        "lds_array_busy_frac": m["SQ_LDS_IDX_ACTIVE"] / cus / cycles,
        "lds_bank_conflict_cycles": m["SQ_LDS_BANK_CONFLICT"],
        "lds_bank_conflict_frac_of_lds_active": m["SQ_LDS_BANK_CONFLICT"] / max(m["SQ_LDS_IDX_ACTIVE"], 1.0),
        "valu_instructions_per_launch": m["SQ_INSTS_VALU"],
        "lds_instructions_per_launch": m["SQ_INSTS_LDS"],
        "waves_resident_per_simd_avg": 4.0 * m["SQ_WAVE_CYCLES"] / simds / cycles,
        "wait_any_frac_of_wave_cycles": m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"],
        "wait_inst_any_frac_of_wave_cycles": m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"],
    }
json.dump({"source": "rocprofv3 --pmc (two SQ passes, scripts/collect_profiles.sh) on `python bench.py --roofline-only`",
           "units": "SQ_ACTIVE_INST_*/SQ_WAVE_CYCLES/SQ_WAIT_* in quad-cycles summed over the chip; GRBM_GUI_ACTIVE summed over 8 XCDs",
           "raw_avg_per_launch": {k: v for k, v in sq.items() if k in derived},
           "derived": derived}, open(os.path.join(OUT, TAG + "_attn_pmc.json"), "w"), indent=1)

# ---- A/B tables, CPU thread sweep, parity report, RCCL log ----------------------------------------------------------------
ab = {}
p = os.path.join(RAW, "ab.txt")
if os.path.exists(p):
    for line in open(p):
        name, _, rest = line.partition(" ")
        try:
            j = json.loads(rest)
            ab[name] = {"ms_per_step": j["ms_per_step"], "images_per_s": j["value"]}
        except Exception:
            pass
    json.dump({"what": "`python bench.py --no-cpu-baseline --no-roofline` (MedT 128, 4 images, one hipGraph replay per step) on ONE "
                       "box, back to back, with one switch flipped (scripts/collect_profiles.sh lists the variables): "
                       "BN_FIN_APPLY_OFF = bn_finalize + bn_apply_act as two launches; BN_CHAN_OFF = bn_act_bwd_stats -> "
                       "bn_bwd_finalize -> bn_bwd_apply instead of one workgroup per (group, channel); WGRAD_R2_CHUNKS = round 2's "
                       "32 chunks of <= 512 positions; CONV_WS_OFF = no wave-split forward / 16384-position dgrad; TWO_PASS_BWD = "
                       "the generic two-pass attention backward; UP2X_SCALAR = one output column per lane; WGRAD_TILE64 = 64x64 instead "
                       "of 32x64 tiles for the recorded weight gradients; VALU_WGRAD = 4x4 "
                       "register tiles; IMMEDIATE = no recorded/grouped launches; ONE_STREAM; NO_SPLIT_FLUSH",
               "runs": ab}, open(os.path.join(OUT, TAG + "_step_ab.json"), "w"), indent=1)
p = os.path.join(RAW, "conv_ab.json")
if os.path.exists(p) and os.path.getsize(p) > 10:
    shutil.copy(p, os.path.join(OUT, TAG + "_conv_ab.json"))
sweep = {}
p = os.path.join(RAW, "cpu_threads.txt")
if os.path.exists(p):
    for line in open(p):
        t, _, rest = line.partition(" ")
        try:
            j = json.loads(rest)
            sweep[t] = {"s_per_step": j["s_per_step"], "images_per_s": j["value"]}
        except Exception:
            pass
    json.dump({"what": "cpu_baseline leg of bench.py (oracle, MedT 128 bs 4, fwd+CE+bwd+Adam, 3 steps after 1 warm-up) on the "
                       "GPU box's host cores at MEDT_CPU_THREADS = 8 / 16 / 32 / 64", "threads": sweep},
              open(os.path.join(OUT, TAG + "_cpu_thread_sweep.json"), "w"), indent=1)
for f, dst in (("parity_report.txt", TAG + "_parity_report.txt"), ("dist_forced_rccl.log", TAG + "_dist_forced_rccl.log"),
               ("graph_host_cost.txt", TAG + "_graph_host_cost.txt"), ("valu_rate.txt", TAG + "_valu_rate_ubench.txt")):
    p = os.path.join(RAW, f)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(OUT, dst))
print(json.dumps({k: v for k, v in derived.items() if "attn_fwd" in k}, indent=1)[:1500])
print("traffic bytes/launch", kern[main]["hbm_bytes_per_launch"])
