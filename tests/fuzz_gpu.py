"""Long randomized parity sweep of the HIP path against the float64 oracle (not collected by pytest: run by hand
on a GPU box, e.g.  python tests/fuzz_gpu.py --cases 600 --seed 7 > gpurun_out/fuzz.log).

Each case draws dtype (bf16 / fp8 e4m3fn / fp32), B, H, Sq, Sk (60 % square), d, causal, layout (dense or
(B,S,H*d) model-layout views), output dtype, a score scale that sometimes forces the optimistic pass to fall back,
and checks O and LSE against an element-wise error bound derived from the kernel's arithmetic (error_bound).  Prints one line per failure and a summary.
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import __graft_entry__ as entry  # noqa: E402

fa = entry.load_package()
import oracle  # noqa: E402  (checker only)

DEV = "cuda:0"
FP8 = getattr(torch, "float8_e4m3fn", None)


def error_bound(in_dtype, out_dtype, ref, ref_abs, smax):
    """Element-wise bound on |O - ref| from the arithmetic the kernel is documented to do (DESIGN.md "Tolerance"):
    ref_abs = sum_k p_k |v_k| (the oracle run on |V|), smax = largest |scale * score| of the problem.
      bf16 / fp8 inputs: every weight is rounded to bf16 before P.V (2^-9 relative) while the normaliser sums the
        unrounded weights (another 2^-9)                                   ->  2^-8 * ref_abs
      fp32 inputs: an fp32 score carries ~smax * 2^-23 * few of rounding noise, which exp() turns into relative
        error of every weight                                              ->  8 * smax * 2^-23 * ref_abs
      fp8 inputs additionally: gfx950's fp8 MFMA (scaled or not) accumulates its products with ~2^-17 relative
        precision (tests/micro/fp8_accumulation_error.py: 5e-6 |score| against 5e-9 for the bf16 MFMA on the same
        values), so every score carries up to 2^-15 * smax                  ->  2 * 2^-15 * smax * ref_abs
      all: 1e-5 absolute; bf16 output adds its own rounding 2^-8 |ref|."""
    if in_dtype == torch.float32:
        b = 1e-5 + (8.0 * max(smax, 4.0) * 2.0 ** -23) * ref_abs + 1e-5 * np.abs(ref)
    else:
        b = 1e-5 + 2.0 ** -8 * ref_abs + 8.0 * max(smax, 4.0) * 2.0 ** -23 * ref_abs
        if in_dtype == FP8:
            b = b + 2.0 * 2.0 ** -15 * smax * ref_abs
    if out_dtype == torch.bfloat16:
        b = b + 2.0 ** -8 * np.abs(ref)
    return b


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    fails, t0, kinds = 0, time.time(), {}
    for i in range(args.cases):
        dtype = [torch.bfloat16, torch.bfloat16, torch.float32, FP8][int(rng.integers(0, 4 if FP8 is not None else 3))]
        if dtype == FP8:
            d = int(rng.choice([128, 128, 64, 96, 32]))
        elif dtype == torch.float32:
            d = int(rng.choice([64, 128, 128, 16, 32, 80, 200, 256]))
        else:
            d = int(rng.choice([64, 128, 128, 128, 80, 96]))
        B, H = int(rng.integers(1, 5)), int(rng.integers(1, 9))
        Sq = int(rng.choice([1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 300, 511, 512, 513, 777, 1024, 1500]))
        Sk = Sq if rng.random() < 0.6 else int(rng.choice([1, 3, 64, 65, 128, 200, 513, 1000, 2048]))
        causal, strided = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        out_dtype = [torch.float32, torch.bfloat16][int(rng.integers(0, 2))]
        boost = float(rng.choice([1.0, 1.0, 1.0, 3.0, 12.0]))         # 12: later tiles exceed tile 0's max by > 2^127
        if B * H * max(Sq, Sk) * d > 6_000_000:
            H = max(1, H // 4)
        g = torch.Generator().manual_seed(int(rng.integers(0, 2**31)))
        mk = lambda S, mul: (torch.randn(B, S, H * d, generator=g) * mul).to(dtype)
        Qm, Km, Vm = mk(Sq, boost), mk(Sk, boost), mk(Sk, 1.0)
        view = lambda t, S: t.view(B, S, H, d).transpose(1, 2)
        Qd, Kd, Vd = view(Qm.to(DEV), Sq), view(Km.to(DEV), Sk), view(Vm.to(DEV), Sk)
        if not strided:
            Qd, Kd, Vd = Qd.contiguous(), Kd.contiguous(), Vd.contiguous()
        O, lse = fa.flash_attention(Qd, Kd, Vd, is_causal=causal, out_dtype=out_dtype, return_lse=True)
        torch.cuda.synchronize()
        f = lambda t, S: view(t, S).float().numpy()
        qn, kn, vn = f(Qm, Sq), f(Km, Sk), f(Vm, Sk)
        ref = oracle.attention_numpy(qn, kn, vn, causal=causal)
        ref_abs = oracle.attention_numpy(qn, kn, np.abs(vn), causal=causal)
        lref = oracle.lse_numpy(qn, kn, causal=causal)
        smax = float(np.abs(qn.astype(np.float64) @ np.swapaxes(kn.astype(np.float64), -1, -2)).max() / np.sqrt(d))
        Oh, lh = O.float().cpu().numpy(), lse.cpu().numpy()
        bad_o = ~np.isfinite(Oh) | (np.abs(Oh - ref) > error_bound(dtype, out_dtype, ref, ref_abs, smax))
        # LSE = (m + log2 l) ln 2 from fp32 scores and an fp32 sum of UNROUNDED weights: score noise + one ulp of |LSE|
        bad_l = np.abs(lh - lref) > 1e-5 + 16.0 * max(smax, 4.0) * 2.0 ** -23 + 2.0 ** -22 * np.abs(lref) + (
            2.0 ** -13 * smax if dtype == FP8 else 0.0)
        key = (str(dtype).split(".")[-1], d in (64, 128))
        kinds[key] = kinds.get(key, 0) + 1
        if bad_o.any() or bad_l.any():
            fails += 1
            print(f"FAIL case {i}: dtype={dtype} B={B} H={H} Sq={Sq} Sk={Sk} d={d} causal={causal} strided={strided} "
                  f"out={out_dtype} boost={boost}: O bad {int(bad_o.sum())}/{bad_o.size} max err {np.nanmax(np.abs(Oh - ref)):.3e}, "
                  f"LSE bad {int(bad_l.sum())}", flush=True)
    print(f"{args.cases} cases, {fails} failed, {time.time() - t0:.1f} s, seed {args.seed}; cases per (dtype, MFMA-path d): {kinds}")
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
