#!/usr/bin/env python3
"""Mint golden vectors by importing the reference's check.py IN THE BUILD CONTAINER.

Run once, here:  python tests/golden/make_golden.py
The reference (/root/reference) never travels to the GPU box; only the data written by this
script (inputs + expected outputs as .npy, plus manifest.json) is committed.

Fixtures:
  F0  cfg0: randn (1,128,64) fp32, H=1           -> output (1,128,64), attn row sums
  F1  check.py:30-38 demo: ones (1,4,8), H=2      -> output == 1, attn == 0.25
  F2  tests/main.cu:24-36: ones S=16 d=16         -> O == 1 (check.py math on (1,16,16), H=1)
  F3  layout: randn (2,64,128), H=2 (d_k=64)      -> pins (B,S,H*d_k) <-> [B,H,S,d]
  F4  causal S=128 d=64: check.py:19-21 with the k>q => -inf mask inserted between :19 and :20
      (check.py has no mask; predicate from kernels/utils.cuh:43 / tests/main.cu:81)
  F5  F0 inputs pre-rounded to bf16 / e4m3fn (torch casts) -> output of check.py on those
  F6  randn (1,256,128) H=1 scaled x3 (peaky softmax, forces online-softmax rescales)
"""
import hashlib
import importlib.util
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/check.py"


def load_check():
    spec = importlib.util.spec_from_file_location("ref_check", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)          # the demo is under __main__, so nothing prints
    return mod


def causal_variant(Q, K, V, num_heads):
    """check.py:11-24 verbatim in structure, with the causal mask between :19 and :20."""
    B, S, dm = Q.shape
    dk = dm // num_heads
    q = Q.view(B, S, num_heads, dk).transpose(1, 2)
    k = K.view(B, S, num_heads, dk).transpose(1, 2)
    v = V.view(B, S, num_heads, dk).transpose(1, 2)
    scores = torch.matmul(q, k.transpose(-2, -1)) / (dk ** 0.5)
    mask = torch.triu(torch.ones(S, S, dtype=torch.bool), diagonal=1)   # key k > query q
    scores = scores.masked_fill(mask, float("-inf"))
    attn = F.softmax(scores, dim=-1)
    out = torch.matmul(attn, v)
    return out.transpose(1, 2).contiguous().view(B, S, dm), attn


def main():
    if not os.path.exists(REF):
        sys.exit("reference not present: golden vectors can only be minted in the build container")
    check = load_check()
    mha = check.multi_head_attention
    manifest = {"generator": "tests/golden/make_golden.py",
                "reference": "check.py:multi_head_attention (imported from /root/reference)",
                "torch": torch.__version__, "fixtures": {}}

    def save(name, arrays, meta):
        entry = dict(meta)
        entry["files"] = {}
        for key, t in arrays.items():
            a = t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
            a = np.ascontiguousarray(a, dtype=np.float32)
            fn = f"{name}_{key}.npy"
            np.save(os.path.join(HERE, fn), a, allow_pickle=False)
            entry["files"][key] = {"file": fn, "shape": list(a.shape), "dtype": "float32",
                                   "sha256": hashlib.sha256(a.tobytes()).hexdigest()}
        manifest["fixtures"][name] = entry

    def randn(seed, *shape):
        g = torch.Generator().manual_seed(seed)
        return torch.randn(*shape, generator=g, dtype=torch.float32)

    # F0
    Q, K, V = randn(100, 1, 128, 64), randn(101, 1, 128, 64), randn(102, 1, 128, 64)
    out, attn = mha(Q, K, V, 1)
    save("F0", {"Q": Q, "K": K, "V": V, "out": out, "attn_rowsum": attn.sum(-1)},
         {"num_heads": 1, "causal": False, "seeds": [100, 101, 102], "layout": "(B,S,H*d_k)"})
    # F1
    one = torch.ones(1, 4, 8)
    out, attn = mha(one, one, one, 2)
    save("F1", {"Q": one, "K": one, "V": one, "out": out, "attn": attn},
         {"num_heads": 2, "causal": False, "layout": "(B,S,H*d_k)"})
    # F2
    one = torch.ones(1, 16, 16)
    out, attn = mha(one, one, one, 1)
    save("F2", {"Q": one, "K": one, "V": one, "out": out},
         {"num_heads": 1, "causal": False, "layout": "(B,S,H*d_k)",
          "note": "tests/main.cu:24-36 inputs; scale 1/sqrt(16) as tests/main.cu:27"})
    # F3
    Q, K, V = randn(300, 2, 64, 128), randn(301, 2, 64, 128), randn(302, 2, 64, 128)
    out, attn = mha(Q, K, V, 2)
    save("F3", {"Q": Q, "K": K, "V": V, "out": out, "attn": attn},
         {"num_heads": 2, "causal": False, "seeds": [300, 301, 302], "layout": "(B,S,H*d_k)"})
    # F4
    Q, K, V = randn(400, 1, 128, 64), randn(401, 1, 128, 64), randn(402, 1, 128, 64)
    out, attn = causal_variant(Q, K, V, 1)
    out_nc, _ = mha(Q, K, V, 1)
    save("F4", {"Q": Q, "K": K, "V": V, "out": out, "out_noncausal": out_nc},
         {"num_heads": 1, "causal": True, "seeds": [400, 401, 402], "layout": "(B,S,H*d_k)",
          "note": "mask k>q inserted between check.py:19 and :20"})
    # F5
    Q, K, V = randn(100, 1, 128, 64), randn(101, 1, 128, 64), randn(102, 1, 128, 64)
    Qb, Kb, Vb = (t.to(torch.bfloat16).float() for t in (Q, K, V))
    out, _ = mha(Qb, Kb, Vb, 1)
    save("F5bf16", {"Q": Qb, "K": Kb, "V": Vb, "out": out},
         {"num_heads": 1, "causal": False, "layout": "(B,S,H*d_k)",
          "note": "F0 inputs rounded to bf16 by torch (RNE)"})
    Q8, K8, V8 = (t.to(torch.float8_e4m3fn).float() for t in (Q, K, V))
    out, _ = mha(Q8, K8, V8, 1)
    save("F5e4m3", {"Q": Q8, "K": K8, "V": V8, "out": out},
         {"num_heads": 1, "causal": False, "layout": "(B,S,H*d_k)",
          "note": "F0 inputs rounded to float8_e4m3fn by torch"})
    # F6
    Q, K, V = 3.0 * randn(600, 1, 256, 128), 3.0 * randn(601, 1, 256, 128), randn(602, 1, 256, 128)
    out, _ = mha(Q, K, V, 1)
    save("F6", {"Q": Q, "K": K, "V": V, "out": out},
         {"num_heads": 1, "causal": False, "seeds": [600, 601, 602], "layout": "(B,S,H*d_k)",
          "note": "Q,K scaled x3: peaky softmax"})

    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print("wrote", len(manifest["fixtures"]), "fixtures to", HERE)


if __name__ == "__main__":
    main()
