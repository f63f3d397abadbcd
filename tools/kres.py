#!/usr/bin/env python3
"""Compact table of hipcc -Rpass-analysis=kernel-resource-usage remarks (stdin): one line per kernel."""
import re, sys
cur = {}
rows = []
for line in sys.stdin:
    m = re.search(r"remark: [^:]*:\d+:\d+: +(Function Name|Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs|LDS Size \[bytes/block\]): +(\S+)", line)
    if not m:
        m = re.search(r"(Function Name|Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs|LDS Size \[bytes/block\]): +(\S+)", line)
        if not m: continue
    k, v = m.group(1), m.group(2)
    if k in ("Function Name", "Name"):
        if cur: rows.append(cur)
        cur = {"name": v}
    else:
        cur[k.split()[0]] = v
if cur: rows.append(cur)
for r in rows:
    n = r["name"]
    n = re.sub(r"^_ZN2fa\d+", "", n)
    n = n.replace("KernelCfgILi", "D").replace("NS_3OptE", "Opt:").replace("vNS_6ParamsE", "")
    print(f"V={r.get('VGPRs','?'):>4} A={r.get('AGPRs','?'):>3} S={r.get('SGPRs','?'):>3} scratch={r.get('ScratchSize','?'):>4} occ={r.get('Occupancy','?')}  {n[:150]}")
