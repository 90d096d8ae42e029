"""profiles/roofline_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of `python bench.py --roofline-only` (both SURVEY.md 8(d)
shapes), stamped with the commit.  Usage (on the GPU box, after the two rocprofv3 --pmc passes):
    python scripts/r5_traffic.py <fetch_dir> <write_dir> <roofline_only_json> <commit> > profiles/roofline_traffic.json
Correction per MI355X_MICROARCH.md (HBM section): gfx950's FETCH_SIZE reports half the bytes of wide coalesced reads -> x2;
WRITE_SIZE as reported; unit KiB."""
import collections, csv, glob, json, os, re, sys


def load(d, counter):
    fs = glob.glob(os.path.join(d, "*", "*_counter_collection.csv")) + glob.glob(os.path.join(d, "*_counter_collection.csv"))
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(max(fs, key=os.path.getsize))):
        if r["Counter_Name"] != counter:
            continue
        k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("medt::", "").replace("(anonymous namespace)::", "").replace("void ", ""))
        agg[k].append(float(r["Counter_Value"]))
    return agg


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
legs = json.load(open(sys.argv[3]))
kern = {}
for k, v in fetch.items():
    if any(t in k for t in ("attn_fwd", "sim_stats", "attn_bwd", "sim_bwd_finalize", "bwd_tables")):
        # the timed loops dominate the launch counts; warm-up / set-up launches of the same kernel see the same shape
        w = write.get(k, [0.0])
        kern[k] = {"FETCH_SIZE_KB_avg": sum(v) / len(v), "WRITE_SIZE_KB_avg": sum(w) / len(w), "launches": len(v),
                   "hbm_bytes_per_launch": int((2 * sum(v) / len(v) + sum(w) / len(w)) * 1024)}


def shape(tag, leg, fwd_pat, sweep_pat, hq):
    fwd = [k for k in kern if re.search(fwd_pat, k)]
    main = max(fwd, key=lambda k: kern[k]["FETCH_SIZE_KB_avg"]) if fwd else None
    parts = {}
    for k in kern:
        if re.search(sweep_pat, k) and "false" in k:
            parts["sweep"] = k
        elif "attn_bwd_fix_kernel<%d>" % hq in k:
            parts["fix"] = k
        elif "attn_bwd_relfix_kernel<%d>" % hq in k:
            parts["relfix"] = k
    out = {"attn_fwd_kernel": main, "attn_fwd_bytes_per_launch": kern[main]["hbm_bytes_per_launch"] if main else None,
           "attn_fwd_algorithmic_bytes": leg["shape"]["bytes_per_launch"], "attn_bwd_kernels": parts,
           "attn_bwd_bytes_per_launch": sum(kern[k]["hbm_bytes_per_launch"] for k in parts.values()) if "sweep" in parts else None,
           "attn_bwd_algorithmic_bytes": leg["bwd_core"]["bytes_per_launch"]}
    if out["attn_fwd_bytes_per_launch"]:
        out["attn_fwd_traffic_over_algorithmic"] = out["attn_fwd_bytes_per_launch"] / out["attn_fwd_algorithmic_bytes"]
    if out["attn_bwd_bytes_per_launch"]:
        out["attn_bwd_traffic_over_algorithmic"] = out["attn_bwd_bytes_per_launch"] / out["attn_bwd_algorithmic_bytes"]
    return out


a = shape("C16_L64", legs["roofline"], r"attn_fwd4r_kernel<1, 64", r"attn_bwd_sweep_kernel<2, 64, 16", 1)
b = shape("C32_L128", legs["also"], r"attn_fwd3_kernel<4, 1, 128", r"attn_bwd_sweep_kernel<4, 128, 32", 2)
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) on `python bench.py --roofline-only` "
                     "(scripts/r5_call2.sh -> scripts/r5_traffic.py)",
           "commit": sys.argv[4],
           "correction": "gfx950: FETCH_SIZE reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section) -> x2; WRITE_SIZE as reported; unit KiB",
           "kernels": kern, **a, "C32_L128": b,
           "note": "attn_fwd traffic = qkv read once + sv|sve written once + row log-sum-exp (excluded from the algorithmic figure by SURVEY.md 8d)"},
          sys.stdout, indent=1)
