#!/usr/bin/env python3
"""Compare the kernels of two hipcc .s files instruction by instruction (comments, directives and symbol names ignored).
usage: isa_compare.py old.s new.s      -- kernels are paired in file order; prints one line per pair."""
import hashlib
import re
import sys


def kernels(path):
    out, name, body = [], None, []
    for line in open(path):
        if re.match(r"^_Z\S+:", line):
            name, body = line.split(":")[0], []
            continue
        if name is None:
            continue
        t = line.strip()
        if t.startswith(".Lfunc_end") or t.startswith(".end_amdhsa_kernel"):
            out.append((name, body))
            name = None
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        t = t.split(";")[0].strip()
        t = re.sub(r"_Z\w+", "SYM", t)
        if t:
            body.append(t)
    return out


a, b = kernels(sys.argv[1]), kernels(sys.argv[2])
print(f"{len(a)} kernels vs {len(b)} kernels")
same = 0
for (na, ba), (nb, bb) in zip(a, b):
    ha, hb = hashlib.sha1("\n".join(ba).encode()).hexdigest()[:12], hashlib.sha1("\n".join(bb).encode()).hexdigest()[:12]
    eq = ha == hb
    same += eq
    if not eq:
        first = next((i for i, (x, y) in enumerate(zip(ba, bb)) if x != y), min(len(ba), len(bb)))
        print(f"DIFF  {len(ba)} vs {len(bb)} instructions, first difference at {first}: {ba[first] if first < len(ba) else None!r} vs {bb[first] if first < len(bb) else None!r}")
print(f"{same} of {min(len(a), len(b))} kernel pairs identical")
