cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3pmc2; rm -rf $O; mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z0-9_]+" | sort -u > $O/sq_counters.txt
for ls in 16 32; do
MEDT_BWD_LS=$ls timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $O/a_ls$ls -- python bench.py --roofline-only > $O/a$ls.log 2>&1
MEDT_BWD_LS=$ls timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d $O/b_ls$ls -- python bench.py --roofline-only > $O/b$ls.log 2>&1
MEDT_BWD_LS=$ls timeout 200 rocprofv3 --pmc SQ_IFETCH SQ_INSTS_SALU SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_THREAD_CYCLES_VALU SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL --kernel-trace --output-format csv -d $O/c_ls$ls -- python bench.py --roofline-only > $O/c$ls.log 2>&1
done
find $O -name "*kernel_trace.csv" -delete
python - <<'PY'
import csv, glob, collections, json
out = {}
for d in sorted(glob.glob('gpurun_out/r3pmc2/*_ls*')):
    tag = d.split('_ls')[-1]
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            n = r['Kernel_Name']
            if 'sweep' in n:
                acc['sweep_ls' + tag + ('_gates' if 'true' in n else '_nogates')][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in acc.items():
            out.setdefault(k, {}).update({c: sum(x) / len(x) for c, x in v.items()})
json.dump(out, open('gpurun_out/r3pmc2/summary.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
PY
wc -l $O/sq_counters.txt
