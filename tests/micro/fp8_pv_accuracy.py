"""What would it cost in accuracy to run P.V of the fp8 path (BASELINE cfg3) on the fp8 MFMA as well?

The fp8 kernel computes QK^T on the block-scaled fp8 MFMA (exact products, fp32 scores) and keeps P.V on the bf16 MFMA
(weights rounded to bf16, V widened exactly).  Putting P.V on the fp8 rate needs the weights as e4m3 (3 mantissa bits:
2^-4 relative rounding, against 2^-9 for bf16), with a per-row power-of-two pre-scale so that p <= 1 uses the format's range.
This script simulates exactly that arithmetic on the CPU (numpy float64 everywhere except the rounding of P), on the same input
distribution bench.py uses for cfg3 and on sharper score distributions, and reports the output error against the exact result
next to the bf16-weights figure.  It needs no GPU; the table it prints is committed as profiles/r02_fp8_pv_accuracy.txt and
quoted in DESIGN.md section 4.

    python tests/micro/fp8_pv_accuracy.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle  # CPU restatement of the reference; here only its rounding helpers (round_e4m3fn, round_bf16)


def attention_with_rounded_weights(q, k, v, scale, rnd):
    s = (q.astype(np.float64) @ k.astype(np.float64).T) * scale
    p = np.exp(s - s.max(-1, keepdims=True))
    pr = rnd(p)
    return (pr @ v.astype(np.float64)) / pr.sum(-1, keepdims=True), (p @ v.astype(np.float64)) / p.sum(-1, keepdims=True)


def rnd_bf16(p):
    return oracle.round_bf16(p.astype(np.float32)).astype(np.float64)


def rnd_e4m3_scaled(p):
    # p <= 1: pre-scale by 2^8 (exact), round to e4m3fn (max 448, subnormals down to 2^-9), undo
    return oracle.round_e4m3fn((p * 256.0).astype(np.float32)).astype(np.float64) / 256.0


def main():
    rng = np.random.default_rng(0)
    d = 128
    print(f"{'case':58s} {'weights':>8s} {'max_abs':>10s} {'rms':>10s} {'pass@4e-3':>10s} {'pass@1e-3':>10s}")
    for S, boost, rows in ((16384, 1.0, 256), (4096, 1.0, 256), (4096, 3.0, 256), (512, 3.0, 256), (64, 1.0, 64)):
        q = oracle.round_e4m3fn((boost * rng.standard_normal((rows, d))).astype(np.float32))
        k = oracle.round_e4m3fn((boost * rng.standard_normal((S, d))).astype(np.float32))
        v = oracle.round_e4m3fn(rng.standard_normal((S, d)).astype(np.float32))
        for name, rnd in (("bf16", rnd_bf16), ("e4m3", rnd_e4m3_scaled)):
            got, ref = attention_with_rounded_weights(q, k, v, 1.0 / np.sqrt(d), rnd)
            err = np.abs(got - ref)
            case = f"S={S}, d={d}, scores ~ N(0, {boost ** 2:.0f}^2), {rows} query rows, V ~ N(0,1) e4m3"
            print(f"{case:58s} {name:>8s} {err.max():10.3e} {np.sqrt((err ** 2).mean()):10.3e} "
                  f"{(err <= 4e-3 + 4e-3 * np.abs(ref)).mean():10.6f} {(err <= 1e-3 + 1e-3 * np.abs(ref)).mean():10.6f}")


if __name__ == "__main__":
    main()
