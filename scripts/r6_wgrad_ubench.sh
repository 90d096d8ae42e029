#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6_wgub; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/p -- python scripts/wgrad_ubench.py > $O/log.txt 2>&1
T=$(ls -S $(find $O/p -name "*kernel_trace.csv") | head -1)
python - "$T" <<'PY'
import csv, sys, statistics
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
cases = ["decoderf 16->16 K3 128x128 N4 (768 blk x 2 steps)", "conv3 128->8 K3 64x64 N4 (288 x 8)", "conv2 8->128 K3 64x64 N4 (128 x 8)", "qkv 16->32 K1 64x64 N4 (16 x 8)", "adjust 16->2 K1 128x128 N4 (256 x 2)", "local 32->64 K1 16x16 N64 (32 x 8)"]
cur = -1; d = {}
for r in rows:
    n = r["Kernel_Name"]
    if "conv_wgrad_mfma_grouped" in n or "reduce_rows_grouped" in n or "channel_sum_grouped" in n:
        key = "grouped" if "wgrad" in n else ("reduce" if "reduce" in n else "csum")
        d.setdefault((cur, key), []).append(((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))))
    # case boundaries: the "case ... done" prints are not in the trace; count distinct grid sizes of the grouped kernel instead
seen = []
out = {}
for (c, key), v in d.items():
    pass
# group grouped-kernel launches by grid size (each case has its own block count)
import collections
byg = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"]
    if "conv_wgrad_mfma_grouped" in n:
        g = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
        byg.setdefault(g, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for g, v in byg.items():
    print("grouped kernel, %5d workgroups: n %3d  median %7.1f us  min %7.1f  max %7.1f" % (g, len(v), statistics.median(v), min(v), max(v)))
PY
rm -rf $O/p; cat $O/log.txt | tail -8
