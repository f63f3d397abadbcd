#!/usr/bin/env python3
"""Launch-to-launch gap of K back-to-back calls on one stream: eager launches against one hipGraph that holds the same K calls
(GPU only).  usage: python3 tools/graph_gap.py [K]"""
import importlib
import sys
import time

import torch

sys.path.insert(0, ".")
fa = importlib.import_module("flash-attention-cuda-c_amd")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B, H, S, d = 8, 16, 4096, 128
dev = "cuda:0"
q = torch.randn(B, H, S, d, device=dev, dtype=torch.bfloat16)
k = torch.randn_like(q)
v = torch.randn_like(q)
o = torch.empty(B, H, S, d, device=dev, dtype=torch.float32)
flops = 2.0 * B * H * S * S * d


def run(n):
    for _ in range(n):
        fa.flash_attention(q, k, v, o, is_causal=True)


for _ in range(200):          # prime the clocks
    run(1)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    run(K)
torch.cuda.synchronize()
for name, f in (("eager", lambda: run(K)), ("graph", g.replay), ("eager", lambda: run(K)), ("graph", g.replay)):
    ts = []
    for _ in range(15):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        f()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / K)
    ts.sort()
    print(f"{name}: K={K} per call median {ts[len(ts)//2]*1e3:.4f} ms  min {ts[0]*1e3:.4f} ms  -> {flops/ts[len(ts)//2]/1e12:.1f} TFLOP/s")
