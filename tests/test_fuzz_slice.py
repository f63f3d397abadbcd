"""A bounded slice of the randomized sweep (tests/fuzz_gpu.py) under `-m gpu`, plus the two fp8 LSE cases that round 1's
long sweeps flagged, pinned by (seed, case number) and reported term by term.

Round 1 answered those two misses ("LSE bad 1": seed 11 case 289, causal, boost 12, tracked-fallback path; seed 99 case 1658,
d = 32 padded, Sk = 65) by widening an ad-hoc `2^-14 * smax` allowance to `2^-13 * smax`.  The bound is now derived:
LSE moves by at most the largest score error among the visible keys, and the fp8 MFMA's score error is EPS_FP8 relative to the
LARGEST product behind the score (bounded by scale * max|q_i| * max|k_i|), with EPS_FP8 taken from a committed measurement of the
instruction itself (tests/unit_kernels "MEASURE fp8 accumulation" -> profiles/r02_unit_kernels.log) -- not tuned on these cases.
(The recurrence whose output the LSE is: /root/reference/kernels/utils.cuh:58-81.)
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import fuzz_gpu as fz  # noqa: E402


def _replay(seed, index):
    for c in fz.draw_cases(seed, index + 1):
        if c["i"] == index:
            return c
    raise AssertionError("case stream too short")


def _report(c, r):
    w = np.unravel_index(np.argmax(r["l_err"] / r["l_bound"]), r["l_err"].shape)
    term = lambda v: float(v[w]) if isinstance(v, np.ndarray) else float(v)
    print(f"\n{fz.describe(c)}\n  max scaled score {r['smax']:.1f}; worst LSE element {w}: observed |LSE - ref| = {r['l_err'][w]:.4e}, "
          f"bound {r['l_bound'][w]:.4e} = " + " + ".join(f"{k} {term(v):.3e}" for k, v in r["l_terms"].items()) +
          f"\n  round-1 ad-hoc allowances at this smax: 2^-14*smax = {2.0 ** -14 * r['smax']:.3e} (failed), 2^-13*smax = {2.0 ** -13 * r['smax']:.3e}"
          f"\n  largest err/bound over all LSE elements {r['worst_l']:.3f}; O: {r['bad_o']} bad, largest err/bound {r['worst_o']:.3f}")


@pytest.mark.skipif(fz.FP8 is None, reason="torch build without float8_e4m3fn")
def test_fp8_lse_pinned_case_seed99_1658():
    """profiles/r01_fuzz_cases.log line 7: fp8, B=3 H=6 Sq=256 Sk=65, d = 32 (padded onto the 128-wide kernel), causal, strided,
    boost 3 -- replayed from the seeded stream."""
    c = _replay(99, 1658)
    expect = dict(B=3, Sq=256, Sk=65, d=32, causal=True, strided=True, boost=3.0)
    assert c["dtype"] == fz.FP8 and all(c[k] == v for k, v in expect.items()), fz.describe(c)   # the logged case, reproduced
    r = fz.run_case(c)
    _report(c, r)
    assert r["bad_l"] == 0 and r["bad_o"] == 0


@pytest.mark.skipif(fz.FP8 is None, reason="torch build without float8_e4m3fn")
@pytest.mark.parametrize("data_seed", [0, 1, 2, 3, 4])
def test_fp8_lse_pinned_case_seed11_289(data_seed):
    """profiles/r01_fuzz_cases.log line 1: fp8, B=4 H=8 Sq=Sk=256, d=128, causal, dense, bf16 output, boost 12 (scores ~ +-700: the
    optimistic pass overflows and the tracked pass recomputes).  That sweep ran an earlier version of the case generator, so the
    stream position no longer reproduces it: the logged PARAMETERS are pinned, on five draws of the data."""
    c = dict(i=289, seed=11, dtype=fz.FP8, B=4, H=8, Sq=256, Sk=256, d=128, causal=True, strided=False, out_dtype=torch.bfloat16,
             boost=12.0, data_seed=data_seed)
    r = fz.run_case(c)
    _report(c, r)
    assert r["bad_l"] == 0 and r["bad_o"] == 0


def test_fuzz_slice():
    """200 cases of stream 2026: every dispatch branch (bf16 / fp8 / fp32 MFMA kernels, generic kernel, padded head dimensions,
    cross lengths, strided layouts, both output types, optimistic pass and its fallback) against the derived bounds."""
    fails, worst_l, worst_o, n_fp8 = [], 0.0, 0.0, 0
    for c in fz.draw_cases(2026, 200):
        if c["dtype"] is None:
            continue
        r = fz.run_case(c)
        n_fp8 += c["dtype"] == fz.FP8
        worst_l, worst_o = max(worst_l, r["worst_l"]), max(worst_o, r["worst_o"])
        if r["bad_o"] or r["bad_l"]:
            fails.append(f"{fz.describe(c)}: O bad {r['bad_o']}/{r['n_o']} (err/bound {r['worst_o']:.2f}), LSE bad {r['bad_l']} (err/bound {r['worst_l']:.2f})")
    print(f"\n200 cases ({n_fp8} fp8): {len(fails)} failed; largest observed error / bound: O {worst_o:.3f}, LSE {worst_l:.3f}")
    assert not fails, "\n".join(fails)
