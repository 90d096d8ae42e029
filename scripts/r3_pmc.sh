# PMC passes over the roofline leg (run ON the GPU box): SQ issue / wait / LDS counters, HBM traffic
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3pmc; rm -rf $O; mkdir -p $O
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq1 -- python bench.py --roofline-only > $O/pmc_sq1.log 2>&1
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq2 -- python bench.py --roofline-only > $O/pmc_sq2.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python bench.py --roofline-only > $O/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python bench.py --roofline-only > $O/pmc_write.log 2>&1
find $O -name "*kernel_trace.csv" -delete
python - <<'PY'
import csv, glob, collections, json
out = {}
for d in sorted(glob.glob('gpurun_out/r3pmc/pmc_*')):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            n = r['Kernel_Name']
            if 'attn_' in n or 'sim_stats' in n:
                acc[n[:70]][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in acc.items():
            out.setdefault(k, {}).update({c: sum(x) / len(x) for c, x in v.items()})
            out[k]['calls'] = len(next(iter(v.values())))
json.dump(out, open('gpurun_out/r3pmc/summary.json', 'w'), indent=1)
print(json.dumps(out, indent=1)[:6000])
PY
