#!/usr/bin/env python3
"""How far from the stated tolerance do bf16 softmax weights put a causal row, as a function of how many keys it sees?

CPU emulation (float64 reference, torch) of the bf16-weights kernels' arithmetic on N(0,1) bf16 inputs, d = 128: exponentials
relative to the row max of key tile 0 (the optimistic pass), fp32 exp, weights rounded to bf16 (RNE), fp32 sum of the UNROUNDED
weights as normaliser.  Prints, per band of query rows (= number of visible keys), the worst element error over all heads as a
fraction of the stated tolerance 1e-3 + 1e-3|ref| (BASELINE.json north_star; the math: /root/reference/check.py:19-21).
This is where FA_EARLY_KEYS = 1024 and the header's "distribution dependence" paragraph come from
(include/flash_attention.h); output of the run behind them: profiles/r04_bf16_weight_error_by_row.txt.
Test / documentation infrastructure: imports nothing of the product.

usage: bf16_weight_error_by_row.py [heads, default 600] [S, default 1536]
"""
import sys
import time

import numpy as np
import torch


def bf16(x):
    return x.to(torch.bfloat16).to(torch.float32)


def main():
    H = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 1536
    d = 128
    bands = [(a, min(a + 256, S)) for a in range(0, S, 256)]
    worst = {b: 0.0 for b in bands}
    mask = torch.triu(torch.ones(S, S, dtype=torch.bool), 1)
    t0 = time.time()
    for h in range(H):
        g = torch.Generator().manual_seed(5000 + h)
        Q, K, V = (bf16(torch.randn(S, d, generator=g)) for _ in range(3))
        s = (Q.double() @ K.double().T) / np.sqrt(d)
        s = s.masked_fill(mask, -float("inf"))
        ref = torch.softmax(s, -1) @ V.double()
        s32 = s.float()
        m0 = s32[:, :64].max(-1, keepdim=True).values          # reference max of the optimistic pass: key tile 0
        p = torch.exp2((s32 - m0) * 1.4426950408889634)
        p = torch.where(mask, torch.zeros_like(p), p)
        O = (bf16(p) @ V) / p.sum(-1, keepdim=True)
        ratio = ((O.double() - ref).abs() / (1e-3 + 1e-3 * ref.abs())).max(-1).values
        for b in bands:
            worst[b] = max(worst[b], float(ratio[b[0]:b[1]].max()))
    print(f"bf16 softmax weights, causal, d = {d}, N(0,1) bf16 inputs, {H} heads x {S} rows ({time.time() - t0:.0f} s)")
    print("rows (= visible keys - 1)   worst |O - ref| / (1e-3 + 1e-3|ref|) over all heads, rows of the band and 128 columns")
    for b in bands:
        print(f"  [{b[0]:5d}, {b[1]:5d})            {worst[b]:.3f}{'   <-- misses the stated tolerance' if worst[b] > 1 else ''}")


if __name__ == "__main__":
    main()
