#!/bin/bash
# Round 5: variants of the attention forward main pass on the roofline shape, ON the GPU box (gpurun), one box, alternating.
# "default" = libmedt_hip.so; every medical-transformer_amd/libmedt_fv_*.so is the same tree built with another -DMEDT_F4R_* setting
# (python -m medt_amd.build style: build(defines=..., lib_path=...)).  HIP-event numbers of bench.py --roofline-only (memset + main
# pass + the repair kernel's early exit), then rocprofv3 kernel statistics of each variant; then the layer tests that run the kernel.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export MEDT_ROOF_ITERS=100
O=gpurun_out/r5_fwd_variants
rm -rf $O && mkdir -p $O
LIBS="default $(ls medical-transformer_amd/libmedt_fv_*.so 2>/dev/null)"
ENVS="${FV_ENVS:-}"            # extra runtime variants of the default library: "name:VAR=value name2:VAR=value"
for rep in 1 2 3; do
  for L in $LIBS; do
    n=$(basename $L .so); n=${n#libmedt_fv_}
    if [ "$L" = default ]; then unset MEDT_LIB_OVERRIDE; else export MEDT_LIB_OVERRIDE=$GRAFT_REPO_ROOT/$L; fi
    timeout 200 python bench.py --roofline-only 2>/dev/null | grep '^{"roofline' | tail -1 > $O/${n}_$rep.json
  done
  unset MEDT_LIB_OVERRIDE
  for E in $ENVS; do
    env ${E#*:} timeout 200 python bench.py --roofline-only 2>/dev/null | grep '^{"roofline' | tail -1 > $O/${E%%:*}_$rep.json
  done
done
for L in $LIBS; do
  n=$(basename $L .so); n=${n#libmedt_fv_}
  if [ "$L" = default ]; then unset MEDT_LIB_OVERRIDE; else export MEDT_LIB_OVERRIDE=$GRAFT_REPO_ROOT/$L; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_$n -- python bench.py --roofline-only > $O/p_$n.log 2>&1
  cp $(ls -S $O/p_$n/*/*_kernel_stats.csv | head -1) $O/kernel_stats_$n.csv; rm -rf $O/p_$n
done
unset MEDT_LIB_OVERRIDE
for E in $ENVS; do
  n=${E%%:*}
  env ${E#*:} timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_$n -- python bench.py --roofline-only > $O/p_$n.log 2>&1
  cp $(ls -S $O/p_$n/*/*_kernel_stats.csv | head -1) $O/kernel_stats_$n.csv; rm -rf $O/p_$n
done
python - <<'PY'
import csv, glob, json, os
O = "gpurun_out/r5_fwd_variants"
out = {}
for f in sorted(glob.glob(O + "/*_[0-9].json")):
    n = os.path.basename(f).rsplit("_", 1)[0]
    try:
        r = json.load(open(f))["roofline"]
        out.setdefault(n, {"launch_ms": [], "frac": []})
        out[n]["launch_ms"].append(round(r["launch_ms"], 5)); out[n]["frac"].append(round(r["frac"], 4))
    except Exception as e:
        out.setdefault(n, {})["error"] = str(e)
for f in sorted(glob.glob(O + "/kernel_stats_*.csv")):
    n = os.path.basename(f)[len("kernel_stats_"):-4]
    for row in csv.DictReader(open(f)):
        if "attn_fwd4r_kernel<1, 64, false" in row["Name"]:
            out.setdefault(n, {})["rocprof_avg_us"] = round(float(row["AverageNs"]) / 1e3, 2)
        if "attn_fwd4r_kernel<1, 64, true" in row["Name"]:
            out.setdefault(n, {})["repair_exit_us"] = round(float(row["AverageNs"]) / 1e3, 2)
json.dump(out, open(O + "/summary.json", "w"), indent=1)
for n, v in out.items():
    print(n, v)
PY
[ -n "$FV_SKIP_TESTS" ] || timeout 600 python -m pytest tests/test_axial_layer_gpu.py -m gpu -q -x -k "four_rows or bound or test_layer_vs_oracle" 2>&1 | tail -2
