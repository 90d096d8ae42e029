#!/bin/bash
# Round 5, GPU call 14: the attention forward main pass with the table pieces carried across chunks (MEDT_F4R_CARRY=2, default build)
# against the previous body (libmedt_carry0.so = the same tree built with -DMEDT_F4R_CARRY=0), on one box, alternating; then the
# layer tests that run the kernel (forced on the small shapes, bound / exact / repair) and the kernel statistics of both.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_call14
rm -rf $O && mkdir -p $O
V0=$GRAFT_REPO_ROOT/medical-transformer_amd/libmedt_carry0.so
for rep in 1 2 3; do
  timeout 200 python bench.py --roofline-only 2>/dev/null | grep '^{"roofline' | tail -1 > $O/carry2_$rep.json
  MEDT_LIB_OVERRIDE=$V0 timeout 200 python bench.py --roofline-only 2>/dev/null | grep '^{"roofline' | tail -1 > $O/carry0_$rep.json
done
python - <<'PY'
import json, glob
for v in ("carry0", "carry2"):
    for f in sorted(glob.glob(f"gpurun_out/r5_call14/{v}_*.json")):
        try:
            j = json.load(open(f)); r = j["roofline"]
            print(v, "frac", r["frac"], "ms", r.get("launch_ms"), "achieved", r["achieved"], "also", [ (a.get("frac"), a.get("launch_ms")) for a in r.get("also", [])])
        except Exception as e:
            print(v, f, "unreadable", e)
PY
timeout 600 python -m pytest tests/test_axial_layer_gpu.py -m gpu -q -x -k "four_rows or bound or test_layer_vs_oracle" 2>&1 | tail -3
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p2 -- python bench.py --roofline-only > $O/p2.log 2>&1
cp $(ls -S $O/p2/*/*_kernel_stats.csv | head -1) $O/roofline_kernel_stats_carry2.csv; rm -rf $O/p2
MEDT_LIB_OVERRIDE=$V0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p0 -- python bench.py --roofline-only > $O/p0.log 2>&1
cp $(ls -S $O/p0/*/*_kernel_stats.csv | head -1) $O/roofline_kernel_stats_carry0.csv; rm -rf $O/p0
grep -h "attn_fwd4r" $O/roofline_kernel_stats_carry*.csv | cut -c1-200
