#!/usr/bin/env python
"""ISA-level issue account of a kernel: instruction classes per basic block (loop nesting from the assembler's comments) of one
function of a `hipcc -save-temps` .s file, priced with the issue rates measured on the MI355X (profiles/r03_valu_rate_ubench.txt).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I csrc -c csrc/axial_fast.hip -save-temps=obj -o /tmp/x.o
    python scripts/isa_account.py /tmp/axial_fast-hip-amdgcn-amd-amdhsa-gfx950.s attn_fwd4r_kernelILi1ELi64ELb0
"""
import collections
import re
import sys

# cycles per wave-instruction per SIMD at 2-4 waves/SIMD (scripts/ubench/valu_rate.hip on the MI355X)
RATE = {"pk_fma": 5.35, "pk_other": 5.35, "fma": 2.6, "valu": 2.6, "exp": 10.5, "dpp": 7.6, "salu": 0.0, "lds": 0.0, "vmem": 0.0, "wait": 0.0,
        "branch": 0.0}


def classify(op):
    if op.startswith("v_pk_fma"):
        return "pk_fma"
    if op.startswith("v_pk_"):
        return "pk_other"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt")):
        return "exp"
    if "_dpp" in op or op.startswith(("v_permlane", "v_readlane", "v_writelane", "v_readfirstlane")):
        return "dpp"
    if op.startswith(("v_fma", "v_fmac", "v_mac")):
        return "fma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith(("s_waitcnt", "s_barrier", "s_nop", "s_sleep")):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
        return "branch"
    return "salu"


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().split(";")[0].rstrip().endswith(":"))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    blocks, cur, depth = [], None, 0
    for l in lines[start + 1:end + 1]:
        t = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            d = re.search(r"Depth=(\d+)", l)
            cur = {"label": m.group(1), "depth": int(d.group(1)) if d else 0, "counts": collections.Counter(), "loophdr": "Loop Header" in l}
            blocks.append(cur)
            continue
        if not t or t.startswith((";", ".", "//")):
            d = re.search(r"Depth=(\d+)", l)
            if d and cur is not None and not cur["counts"]:
                cur["depth"] = int(d.group(1))
            continue
        if cur is None:
            cur = {"label": "entry", "depth": 0, "counts": collections.Counter(), "loophdr": False}
            blocks.append(cur)
        cur["counts"][classify(t.split()[0])] += 1
    tot = collections.Counter()
    print(f"{'block':12s} {'depth':>5s} {'instrs':>7s} {'pk_fma':>7s} {'pk_oth':>7s} {'fma':>5s} {'valu':>5s} {'exp':>5s} {'dpp':>5s} {'lds':>5s} {'vmem':>5s} {'salu':>5s} {'wait':>5s}  issue cycles")
    for b in blocks:
        c = b["counts"]
        n = sum(c.values())
        if n < 12:
            continue
        cyc = sum(RATE[k] * v for k, v in c.items())
        print(f"{b['label']:12s} {b['depth']:5d} {n:7d} {c['pk_fma']:7d} {c['pk_other']:7d} {c['fma']:5d} {c['valu']:5d} {c['exp']:5d} {c['dpp']:5d} "
              f"{c['lds']:5d} {c['vmem']:5d} {c['salu']:5d} {c['wait']:5d}  {cyc:9.0f}")
        tot.update(c)
    print("function total:", dict(tot))


if __name__ == "__main__":
    main()
