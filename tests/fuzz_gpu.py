"""Randomized parity sweep of the HIP path against the float64 oracle.

As a module (tests/test_fuzz_slice.py): `draw_cases(seed, n)` replays the seeded case stream, `run_case(c)` runs one case on
the GPU and returns the observed errors next to every term of the bound they are held to.
As a script (long sweeps, by hand on a GPU box):  python tests/fuzz_gpu.py --cases 600 --seed 7 > gpurun_out/fuzz.log

Every case also flips two coins (from a side generator, so the logged case numbers of earlier rounds still replay): whether the call
asks for the LSE -- without it, non-causal bf16 inputs run the instantiation bench.py times (row sums by MFMA) -- and the softmax-weight
precision (library default / bf16 everywhere / fp16 everywhere).

Each case draws dtype (bf16 / fp8 e4m3fn / fp32), B, H, Sq, Sk (60 % square), d, causal, layout (dense or (B,S,H*d)
model-layout views), output dtype and a score scale that sometimes forces the optimistic pass to fall back, and checks O and
LSE element-wise against a bound derived from the arithmetic the kernel is documented to do (DESIGN.md "Tolerance").

The bound, term by term (ref_abs = sum_k p_k |v_k| = the oracle run on |V|; smax = largest |scale * score|):
  O, bf16 / fp8 inputs   weights rounded to bf16 before P.V (half an ulp of 8 significant bits: <= 2^-8 relative each)  2^-8 ref_abs
  O, bf16, no mask, no LSE request: the normaliser is the MFMA sum of the ROUNDED weights (computers16.hip.h)         + 2^-8 |ref|
  O, fp32 inputs         fp32 score noise through exp()                                              8 smax 2^-23 ref_abs
  O, fp8 inputs, extra   the fp8 MFMA's summation error is EPS_FP8 relative to the LARGEST product of the dot product (MEASURED:
                         tests/unit_kernels "MEASURE fp8 accumulation", profiles/r02_unit_kernels.log): score error <= EPS_FP8 * M,
                         M[q][k] = scale * max_i|q_i| * max_i|k_i| >= scale * max_i|q_i k_i|             2 EPS_FP8 Mmax ref_abs
  O, all                 1e-5 absolute; bf16 output adds 2^-8 |ref|
  LSE                    fp32 scores, fp32 sum of unrounded weights: 1e-5 + 16 smax 2^-23 + 2^-22 |LSE|
  LSE, fp8 inputs, extra |dLSE| <= max over the visible keys of the score error                      EPS_FP8 * max_k M[q][k]
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

# Precision of the fp8 MFMA's internal summation.  tests/unit_kernels measures, over 40 random 32x32x128 products per input scale
# (two chained 32x32x64 MX instructions, as the kernel issues them; profiles/r02_unit_kernels.log):
#   max |D - exact| / max_k|a_k b_k| = 2^-11.6 .. 2^-11.9 with 128 non-zero terms, 2^-12.1 .. 2^-12.4 with 32 (a padded head dimension)
#   -- nearly independent of the term count, while against sum_k|a_k b_k| the same errors read 2^-15.6 and 2^-14.2: the instruction
#   aligns its products to the largest one and keeps ~12 bits below it.  (The bf16 MFMA on the same values: < 2^-24 of sum|ab|.)
# The bound uses twice the largest measured figure.
EPS_FP8 = 2.0 ** -10.6


def draw_cases(seed, n):
    """The first n cases of stream `seed` (the draw order is part of the format: logged case numbers stay reproducible)."""
    rng = np.random.default_rng(seed)
    for i in range(n):
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
        data_seed = int(rng.integers(0, 2**31))
        # Later additions draw from a generator of their own, keyed on (seed, i): the main stream above -- and with it every
        # logged (seed, case number) -- stays what it was.
        #   lse      the call also asks for the LSE or not: for bf16 inputs without the mask these are DIFFERENT instantiations (row
        #            sums by MFMA over the rounded weights / by v_add_f32 over the unrounded ones); the no-LSE one is what bench.py times
        #   weights  None = library default (fp16 weights on rows that see < 1024 keys), or one precision on every row
        rng2 = np.random.default_rng([seed, i, 20261004])
        lse = bool(rng2.integers(0, 2))
        weights = [None, None, torch.bfloat16, torch.float16][int(rng2.integers(0, 4))]
        if dtype != torch.bfloat16 or (weights == torch.float16 and d not in (64, 128)):
            weights = None
        yield dict(i=i, seed=seed, dtype=dtype, B=B, H=H, Sq=Sq, Sk=Sk, d=d, causal=causal, strided=strided,
                   out_dtype=out_dtype, boost=boost, data_seed=data_seed, lse=lse, weights=weights)


def describe(c):
    return (f"seed {c['seed']} case {c['i']}: dtype={c['dtype']} B={c['B']} H={c['H']} Sq={c['Sq']} Sk={c['Sk']} d={c['d']} "
            f"causal={c['causal']} strided={c['strided']} out={c['out_dtype']} boost={c['boost']} lse={c.get('lse', True)} weights={c.get('weights')}")


def run_case(c):
    """Run one case; returns the observed errors and every term of the bounds (numpy arrays / floats)."""
    dtype, B, H, Sq, Sk, d = c["dtype"], c["B"], c["H"], c["Sq"], c["Sk"], c["d"]
    g = torch.Generator().manual_seed(c["data_seed"])
    mk = lambda S, mul: (torch.randn(B, S, H * d, generator=g) * mul).to(dtype)
    Qm, Km, Vm = mk(Sq, c["boost"]), mk(Sk, c["boost"]), mk(Sk, 1.0)
    view = lambda t, S: t.view(B, S, H, d).transpose(1, 2)
    Qd, Kd, Vd = view(Qm.to(DEV), Sq), view(Km.to(DEV), Sk), view(Vm.to(DEV), Sk)
    if not c["strided"]:
        Qd, Kd, Vd = Qd.contiguous(), Kd.contiguous(), Vd.contiguous()
    want_lse = c.get("lse", True)
    res = fa.flash_attention(Qd, Kd, Vd, is_causal=c["causal"], out_dtype=c["out_dtype"], return_lse=want_lse, weights_dtype=c.get("weights"))
    O, lse = res if want_lse else (res, None)
    torch.cuda.synchronize()
    f = lambda t, S: view(t, S).float().numpy()
    qn, kn, vn = f(Qm, Sq), f(Km, Sk), f(Vm, Sk)
    ref = oracle.attention_numpy(qn, kn, vn, causal=c["causal"])
    ref_abs = oracle.attention_numpy(qn, kn, np.abs(vn), causal=c["causal"])
    lref = oracle.lse_numpy(qn, kn, causal=c["causal"])
    scale = 1.0 / np.sqrt(d)
    s = (qn.astype(np.float64) @ np.swapaxes(kn.astype(np.float64), -1, -2)) * scale
    smax = float(np.abs(s).max())
    Oh, lh = O.float().cpu().numpy(), (lse.cpu().numpy() if want_lse else lref.astype(np.float32))   # (no LSE asked for: nothing to compare)
    fp8 = dtype == FP8
    # ---- O ----
    if dtype == torch.float32:
        o_terms = {"abs": 1e-5, "fp32_score_noise": (8.0 * max(smax, 4.0) * 2.0 ** -23) * ref_abs, "ref_ulp": 1e-5 * np.abs(ref)}
    else:
        o_terms = {"abs": 1e-5, "bf16_weights": 2.0 ** -8 * ref_abs, "fp32_score_noise": 8.0 * max(smax, 4.0) * 2.0 ** -23 * ref_abs}
    m_rowmax = None
    if fp8:
        qmax, kmax = np.abs(qn).max(-1).astype(np.float64), np.abs(kn).max(-1).astype(np.float64)      # [B, H, Sq], [B, H, Sk]
        if c["causal"]:                                        # query q sees keys 0 .. min(q, Sk-1)
            kvis = np.maximum.accumulate(kmax, axis=-1)[..., np.minimum(np.arange(Sq), Sk - 1)]
        else:
            kvis = np.broadcast_to(kmax.max(-1, keepdims=True), qmax.shape)
        m_rowmax = scale * qmax * kvis                         # [B, H, Sq]: max over the visible keys of M[q][k]
        o_terms["fp8_accumulation"] = 2.0 * EPS_FP8 * float(m_rowmax.max()) * ref_abs
    if dtype == torch.bfloat16 and not c["causal"] and not want_lse:
        # the instantiation without the mask and without an LSE request normalises by the MFMA sum of the ROUNDED weights: the
        # denominator carries the same 2^-9-per-weight rounding as the numerator (first exercised by the `lse` coin: round 3)
        o_terms["bf16_weights_in_the_normaliser"] = 2.0 ** -8 * np.abs(ref)
    if c["out_dtype"] == torch.bfloat16:
        o_terms["bf16_output"] = 2.0 ** -8 * np.abs(ref)
    o_bound = sum(o_terms.values())
    o_err = np.abs(Oh - ref)
    bad_o = ~np.isfinite(Oh) | (o_err > o_bound)
    # ---- LSE ----
    l_terms = {"abs": 1e-5, "fp32_score_noise": 16.0 * max(smax, 4.0) * 2.0 ** -23, "lse_ulp": 2.0 ** -22 * np.abs(lref)}
    if fp8:
        l_terms["fp8_accumulation"] = EPS_FP8 * m_rowmax
    l_bound = sum(l_terms.values())
    l_err = np.abs(lh - lref)
    bad_l = ~np.isfinite(lh) | (l_err > l_bound)
    return dict(bad_o=int(bad_o.sum()), n_o=bad_o.size, o_err_max=float(np.nanmax(o_err)), bad_l=int(bad_l.sum()), n_l=bad_l.size,
                l_err=l_err, l_bound=l_bound, l_terms=l_terms, o_terms=o_terms, smax=smax,
                worst_l=float((l_err / l_bound).max()), worst_o=float(np.nanmax(o_err / o_bound)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    fails, t0, kinds = 0, time.time(), {}
    worst_fp8_lse, pair_cases = 0.0, 0
    for c in draw_cases(args.seed, args.cases):
        r = run_case(c)
        key = (str(c["dtype"]).split(".")[-1], c["d"] in (64, 128))
        kinds[key] = kinds.get(key, 0) + 1
        if c["dtype"] == torch.bfloat16 and fa.plan(c["B"], c["H"], c["Sq"], c["d"], c["causal"], fa.FA_DTYPE_BF16, fa.FA_DTYPE_F32)["threads"] == 256:
            pair_cases += 1      # (the small-causal-problem kernel at d = 64: fwd_mfma_pair_kernel)
        if c["dtype"] == FP8:
            worst_fp8_lse = max(worst_fp8_lse, r["worst_l"])
        if r["bad_o"] or r["bad_l"]:
            fails += 1
            print(f"FAIL {describe(c)}: O bad {r['bad_o']}/{r['n_o']} max err {r['o_err_max']:.3e} (worst err/bound {r['worst_o']:.2f}), "
                  f"LSE bad {r['bad_l']} (worst err/bound {r['worst_l']:.2f})", flush=True)
    print(f"{args.cases} cases, {fails} failed, {time.time() - t0:.1f} s, seed {args.seed}; fp8 LSE worst err/bound {worst_fp8_lse:.3f}; "
          f"cases per (dtype, MFMA-path d): {kinds}; through the pair kernel: {pair_cases}")
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
