#!/usr/bin/env python
"""Dependent chains of ONE replayed training step from a rocprofv3 kernel trace: the step's launches split by HIP queue
and by the loss kernel (ce_fwd) into global-forward / local-forward / local-backward / global-backward, with the busy time
and launch count of each chain and its largest kernels.  (The profiler serialises the queues: durations are per kernel.)

    python scripts/step_chains.py TRACE_kernel_trace.csv [out.json|-] [rows]"""
import collections
import csv
import json
import re
import sys


def short(n):
    n = n.replace("medt::", "").replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"[<(].*", "", n)


def main():
    rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
    ad = [i for i, r in enumerate(rows) if "adam_step" in r["Kernel_Name"]]
    step = rows[ad[-3] + 1:ad[-2] + 1]
    ce = next(i for i, r in enumerate(step) if "ce_fwd" in r["Kernel_Name"])
    wf = next(r["Queue_Id"] for r in step if "wopos_small_fwd" in r["Kernel_Name"] or "patch_gather" in r["Kernel_Name"])
    chains = collections.OrderedDict((k, []) for k in ("global_fwd", "local_fwd", "local_bwd", "global_bwd"))
    for i, r in enumerate(step):
        loc_q = r["Queue_Id"] == wf
        if i < ce:
            chains["local_fwd" if loc_q else "global_fwd"].append(r)
        else:
            chains["global_bwd" if loc_q else "local_bwd"].append(r)
    out = {"launches": len(step), "chains": {}}
    for name, rs in chains.items():
        per = collections.OrderedDict()
        for r in rs:
            e = per.setdefault(short(r["Kernel_Name"]), [0, 0.0])
            e[0] += 1
            e[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        busy = sum(v[1] for v in per.values())
        out["chains"][name] = {"launches": len(rs), "busy_us": round(busy, 1),
                               "kernels": {k: {"launches": v[0], "busy_us": round(v[1], 1)} for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])}}
        print(f"{name}: {len(rs)} launches, {busy:.0f} us busy")
        for k, v in list(out["chains"][name]["kernels"].items())[:int(sys.argv[3]) if len(sys.argv) > 3 else 12]:
            print(f"    {v['launches']:3d} x {k:42s} {v['busy_us']:8.1f} us")
    if len(sys.argv) > 2 and sys.argv[2] != "-":
        json.dump(out, open(sys.argv[2], "w"), indent=0)


if __name__ == "__main__":
    main()
