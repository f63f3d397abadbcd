#!/usr/bin/env python3
"""Per-basic-block instruction census of one kernel in a hipcc .s file (tuning aid).
usage: isa_blocks.py file.s <substring of the kernel symbol> [min_mfma]"""
import re, sys, collections
path, key = sys.argv[1], sys.argv[2]
min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 1
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().endswith(l.split(":")[0][-10:]) or (l.startswith("_Z") and key in l and ":" in l))
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith("\t.end_amdhsa_kernel") or lines[i].startswith(".Lfunc_end"))
def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("ds_read") or op.startswith("ds_load"): return "ds_read"
    if op.startswith("ds_write") or op.startswith("ds_store"): return "ds_write"
    if op.startswith("buffer_load") or op.startswith("global_load"): return "vmem_ld"
    if op.startswith("buffer_store") or op.startswith("global_store"): return "vmem_st"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("v_exp") or op.startswith("v_log") or op.startswith("v_rcp"): return "trans"
    if op.startswith("v_accvgpr"): return "accvgpr"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_"): return "salu"
    return "other"
blocks, cur, name = [], collections.Counter(), "entry"
ops = collections.Counter()
for l in lines[start + 1:end]:
    m = re.match(r"^(\.LBB[0-9_]+):", l)
    if m:
        blocks.append((name, cur, ops)); cur, ops, name = collections.Counter(), collections.Counter(), m.group(1); continue
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."): continue
    op = t.split()[0]
    cur[cls(op)] += 1
    ops[op] += 1
blocks.append((name, cur, ops))
for name, c, ops in blocks:
    if c["mfma"] >= min_mfma:
        tot = sum(c.values())
        print(f"{name}: total {tot}  " + "  ".join(f"{k}={v}" for k, v in sorted(c.items(), key=lambda x: -x[1])))
        if "-v" in sys.argv:
            print("    " + "  ".join(f"{k}:{v}" for k, v in sorted(ops.items(), key=lambda x: -x[1])[:40]))
