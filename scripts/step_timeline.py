#!/usr/bin/env python
"""Per-queue launch sequence of ONE replayed training step from a rocprofv3 kernel trace.

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python bench.py --no-cpu-baseline --no-roofline
    python scripts/step_timeline.py DIR/**/*_kernel_trace.csv [out.json]

The step is the launches between two consecutive adam_step kernels late in the trace.  The profiler serialises the HIP
queues, so durations are per kernel (busy time), not wall time; what the table shows is which launches sit on which
dependent chain (queue = branch) and what each costs."""
import collections
import csv
import json
import re
import sys


def short(n):
    n = n.replace("medt::", "").replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", n)


def main():
    rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
    ad = [i for i, r in enumerate(rows) if "adam_step" in r["Kernel_Name"]]
    step = rows[ad[-3] + 1:ad[-2] + 1]
    queues = collections.OrderedDict()
    for r in step:
        queues.setdefault(r["Queue_Id"], []).append(r)
    out = {"launches": len(step), "queues": {}}
    for q, rs in queues.items():
        busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs) / 1e3
        per = collections.OrderedDict()
        for r in rs:
            k = re.sub(r"<.*", "", short(r["Kernel_Name"]))
            e = per.setdefault(k, [0, 0.0])
            e[0] += 1
            e[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        out["queues"][q] = {"launches": len(rs), "busy_us": round(busy, 1),
                            "kernels": {k: {"launches": v[0], "busy_us": round(v[1], 1)} for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])},
                            "sequence": [[short(r["Kernel_Name"])[:70], round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1),
                                          int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) //
                                          max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])),
                                          int(r["Workgroup_Size_X"]), int(r["VGPR_Count"]), int(r["LDS_Block_Size"])] for r in rs]}
        print(f"queue {q}: {len(rs)} launches, {busy:.0f} us busy")
        for k, v in out["queues"][q]["kernels"].items():
            print(f"    {v['launches']:3d} x {k:45s} {v['busy_us']:8.1f} us")
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=0)


if __name__ == "__main__":
    main()
