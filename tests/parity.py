"""Parity figures at the tolerance BASELINE.json / BASELINE.md section 4 state: |O - ref| <= 1e-3 + 1e-3*|ref|
(`north_star`: "O matching check.py to rtol=1e-3"; the math being matched: /root/reference/check.py:19-21).

Test infrastructure (and bench.py's cpu_baseline leg): given the HIP path's output and the oracle's output on the
same inputs, report max-abs error, max-rel error and the FRACTION of elements inside the stated tolerance, so a path
that cannot meet it element-wise (bf16 weights: 2^-9 relative per weight) says by how much instead of silently
asserting a looser bound.
"""
import numpy as np

STATED_ATOL = 1e-3
STATED_RTOL = 1e-3


def parity_report(got, ref, atol=STATED_ATOL, rtol=STATED_RTOL):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    err = np.abs(got - ref)
    finite = np.isfinite(err)
    ok = finite & (err <= atol + rtol * np.abs(ref))
    denom = np.maximum(np.abs(ref), 1e-30)
    e = np.where(finite, err, np.inf)
    return {
        "n": int(err.size),
        "max_abs_err": float(e.max()) if err.size else 0.0,
        # relative error where the reference is not tiny (|ref| >= atol): below that the absolute term rules
        "max_rel_err": float((e / denom)[np.abs(ref) >= atol].max()) if (np.abs(ref) >= atol).any() else 0.0,
        "rms_err": float(np.sqrt(np.mean(np.where(finite, err, 0.0) ** 2))) if err.size else 0.0,
        "pass_frac_at_1e-3": float(ok.mean()) if err.size else 1.0,
        "atol": atol, "rtol": rtol,
    }
